"""Write the synthetic BASELINE configs[1] workload to disk as FASTA (genome) + FASTQ (reads, description
pos=<window start>), the on-disk formats of SURVEY.md section 8f rank 2:

  python scripts/make_synth_fastx.py out_dir [n_reads] [window] [read_len]
  python bench.py --dataset fastx --fasta out_dir/genome.fa.gz --fastq out_dir/reads.fq.gz --pairs 2000000
"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from mgl_amd import formats, synth

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
window = int(sys.argv[3]) if len(sys.argv) > 3 else 256
read_len = int(sys.argv[4]) if len(sys.argv) > 4 else 150
os.makedirs(out, exist_ok=True)
genome, win, reads = synth.window_batch(42, n, window=window, read_len=read_len, genome_len=1 << 22)
formats.write_fasta(os.path.join(out, "genome.fa.gz"), [("synth", "seed=42 len=%d" % len(genome), genome.tobytes())])
formats.write_fastq(os.path.join(out, "reads.fq.gz"),
                    (("r%d" % k, "pos=%d" % int(win[k]), reads[k].tobytes(), None) for k in range(n)))
print("wrote", out, n, "reads")
