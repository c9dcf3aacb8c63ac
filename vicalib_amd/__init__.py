"""vicalib_amd: MI355X-native solver core for arpg/vicalib's calibration hot path.

The product is the C-ABI library built from vicalib_amd/csrc (see include/vicalib_amd.h);
this package only carries the ctypes loader used by tests/bench and the synthetic
problem generator.  It never imports anything from oracle/.
"""
