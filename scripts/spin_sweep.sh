for slots in 0 8 14 15 16 20 32; do for t in 16 32 64; do echo -n "slots=$slots "; MGL_SW_COALESCE_SPIN_SLOTS=$slots timeout -k 10 120 tests/cpp/coalesce_bench $t 3000 50; done; done
