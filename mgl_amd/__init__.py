"""mgl_amd -- MI355X-native drop-in for mgl's Smith-Waterman affine-gap alignment core.

The package holds only what that hot path (and the "next" rows of SURVEY.md section 8f) needs:

* ``csrc/``            HIP kernels and the two C-ABI libraries (``libmgl_sw_hip.so``, ``libmgl_pairhmm_hip.so``)
* ``_lib``             ctypes loader for include/mgl_sw.h
* ``smithwaterman``    mirror of the reference's Java operator (MicrosoftSmithWaterman.load / align / close) + batches
* ``device_batch``     HBM-resident batches (ASCII, 2-bit packed, geometry-grouped) as used by bench.py
* ``formats``          FASTA / FASTQ / BAM readers, CIGAR text <-> BAM binary
* ``dist``             one-process-per-GPU sharding and the RCCL score gather
* ``synth``            seeded synthetic workloads
* ``pairhmm``          mirror of MicrosoftPairHmm over include/mgl_pairhmm.h
* ``protein``          substitution-matrix scoring helpers (BLOSUM62; extension, no reference path)
"""
__version__ = "0.1.0"
