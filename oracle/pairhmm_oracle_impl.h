/* pairhmm_oracle_impl.h -- body of the restatement, included once per precision (PHO_T = float / double).
 * TEST INFRASTRUCTURE ONLY (see pairhmm_oracle.h). */

static PHO_T PHO_SUFFIX(g_ph2pr)[128];
static PHO_T PHO_SUFFIX(g_jacobian)[JAC_SIZE];
static PHO_T PHO_SUFFIX(g_m2m)[M2M_SIZE];

/* Context.h:57-59 */
static int PHO_SUFFIX(fast_round)(PHO_T d) { return (d > (PHO_T)0.0) ? (int)(d + (PHO_T)0.5) : (int)(d - (PHO_T)0.5); }

/* Context.h:61-86 */
static PHO_T PHO_SUFFIX(approx_log10_sum)(PHO_T small, PHO_T big)
{
    if (small > big) {
        const PHO_T t = big;
        big = small;
        small = t;
    }
    if (isinf(small) || isinf(big)) return big;
    const PHO_T diff = big - small;
    if (diff >= (PHO_T)JAC_TOL) return big;
    const int ind = PHO_SUFFIX(fast_round)((PHO_T)(diff * ((PHO_T)JAC_INV_STEP)));
    return big + PHO_SUFFIX(g_jacobian)[ind];
}

static void PHO_SUFFIX(init_context)(void)
{
    /* Context.h:41-46 (Jacobian first, :27-32) */
    for (int k = 0; k < JAC_SIZE; k++)
        PHO_SUFFIX(g_jacobian)[k] = (PHO_T)(log10(1.0 + pow(10.0, -((double)k) * JAC_STEP)));
    /* Context.h:49-59 */
    const double LN10 = log(10);
    const double INV_LN10 = 1.0 / LN10;
    for (int i = 0, offset = 0; i <= MAX_QUAL; offset += ++i)
        for (int j = 0; j <= i; j++) {
            const double log10Sum = PHO_SUFFIX(approx_log10_sum)(-0.1f * i, -0.1f * j);
            const double m2mLog10 = log1p(-fmin(1.0, pow(10, log10Sum))) * INV_LN10;
            PHO_SUFFIX(g_m2m)[offset + j] = (PHO_T)(pow(10, m2mLog10));
        }
    for (int x = 0; x < 128; x++) PHO_SUFFIX(g_ph2pr)[x] = PHO_POW10_NEG_TENTH(x);
}

/* Context.h:121-133 / 160-172: both qualities are < 128 here (callers mask with 127), so the table branch */
static PHO_T PHO_SUFFIX(mm_prob)(int insQual, int delQual)
{
    int minQual = delQual, maxQual = insQual;
    if (insQual <= delQual) {
        minQual = insQual;
        maxQual = delQual;
    }
    return PHO_SUFFIX(g_m2m)[((maxQual * (maxQual + 1)) >> 1) + minQual];
}

/* compute_prob_scalar.cc:47-342 as a plain row-major pass.  The reference walks anti-diagonals with three
 * rotating arrays; the value of every cell and the order of the final sum (ascending column, :214,:318) are
 * the same.  Cell update :36-43; per-row tables :68-86; row 0 / column 0 :122-136,:150-152. */
static PHO_T PHO_SUFFIX(forward)(const pho_read *read, const uint8_t *hap, int32_t hap_len)
{
    const int ROWS = read->len + 1, COLS = hap_len + 1;
    const PHO_T threeOver = (PHO_T)1.0 / (PHO_T)3.0; /* :18 */
    PHO_T *pMM = (PHO_T *)malloc(sizeof(PHO_T) * (size_t)ROWS * 6);
    PHO_T *pGapM = pMM + ROWS, *pMX = pGapM + ROWS, *pMY = pMX + ROWS, *pZZ = pMY + ROWS, *Distm = pZZ + ROWS;
    PHO_T *rows = (PHO_T *)malloc(sizeof(PHO_T) * (size_t)COLS * 6);
    PHO_T *M0 = rows, *X0 = M0 + COLS, *Y0 = X0 + COLS, *M1 = Y0 + COLS, *X1 = M1 + COLS, *Y1 = X1 + COLS;
    for (int r = 1; r < ROWS; r++) {
        const int _i = read->ins[r - 1] & 127, _d = read->del[r - 1] & 127, _c = read->gcp[r - 1] & 127;
        pMM[r] = PHO_SUFFIX(mm_prob)(_i, _d);
        pGapM[r] = (PHO_T)1.0 - PHO_SUFFIX(g_ph2pr)[_c];
        pMX[r] = PHO_SUFFIX(g_ph2pr)[_i];
        pMY[r] = PHO_SUFFIX(g_ph2pr)[_d];
        pZZ[r] = PHO_SUFFIX(g_ph2pr)[_c];
        Distm[r] = PHO_SUFFIX(g_ph2pr)[read->qual[r - 1] & 127];
    }
    const PHO_T yInitial = PHO_INITIAL / hap_len; /* :101 */
    for (int c = 0; c < COLS; c++) {
        M0[c] = (PHO_T)0.0;
        X0[c] = (PHO_T)0.0;
        Y0[c] = yInitial;
    }
    for (int r = 1; r < ROWS; r++) {
        M1[0] = X1[0] = Y1[0] = (PHO_T)0.0; /* column 0 of rows >= 1, :150-152 */
        const uint8_t _rs = read->bases[r - 1];
        for (int c = 1; c < COLS; c++) {
            const uint8_t _hap = hap[c - 1];
            const int bMatch = (_rs == _hap) | (_rs == 'N') | (_hap == 'N'); /* :27 */
            PHO_T distm = Distm[r];
            if (bMatch)
                distm = (PHO_T)1.0 - distm;
            else
                distm = distm * threeOver;
            M1[c] = distm * (M0[c - 1] * pMM[r] + (X0[c - 1] + Y0[c - 1]) * pGapM[r]); /* :39 */
            Y1[c] = M1[c - 1] * pMY[r] + Y1[c - 1] * pZZ[r];                            /* :41 */
            X1[c] = M0[c] * pMX[r] + X0[c] * pZZ[r];                                    /* :43 */
        }
        PHO_T *t;
        t = M0; M0 = M1; M1 = t;
        t = X0; X0 = X1; X1 = t;
        t = Y0; Y0 = Y1; Y1 = t;
    }
    PHO_T result = (PHO_T)0.0;
    for (int c = 1; c < COLS; c++) result += M0[c] + X0[c]; /* :214,:318 */
    free(rows);
    free(pMM);
    return result;
}
