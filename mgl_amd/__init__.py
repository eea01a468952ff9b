"""mgl_amd -- MI355X-native drop-in for mgl's Smith-Waterman affine-gap alignment core.

The package holds only what that one hot path needs: ``csrc/`` (HIP kernels and the
C-ABI library ``libmgl_sw_hip.so``), a ctypes host binding that mirrors the reference's
operator interface, the multi-GPU sharding helper and the synthetic workload generators.
"""
__version__ = "0.1.0"
