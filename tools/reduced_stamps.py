"""Phase timeline of k_reduced on BASELINE cfg3 (final stage): build the library with -DVC_REDUCED_STAMPS and point VICALIB_AMD_LIB
at it.  Stamps are shader-clock cycles (__builtin_readcyclecounter: the clock the wavefront runs at, ~2.4 GHz under load): printed as
cycles and as shares of the kernel; scale by the kernel's duration from a trace for microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
base = synth.BASELINE_CONFIGS[name]
if len(sys.argv) > 2:       # frame count override: the per-rank size of a multi-GPU configuration
    base = synth.Config(models=base.models, grid=base.grid, n_frames=int(sys.argv[2]), imu=base.imu, extrinsics_prior=base.extrinsics_prior)
p = synth.generate_native(base)
cal = ViCalibrator(0).load_problem(p)
if p.imu_t is not None:
    cal.SetStageLimit(3); cal.Solve()
else:
    cal.SetCalibrateImu(False)
cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
names = ["entry", "partials summed", "costs summed", "camera blocks", "S complete / solve starts", "solve done", "tail done"]
deferred = not (0 <= st[6] - st[5] < 1e7)       # (round 6: the tail rides in the back-substitution's launch -- stamp 6 is never written)
if deferred: st[6] = st[7]; names[6] = "kernel's end (step + IMU parameters stored, flag raised)"
tot = st[6] - st[0]
for i in range(1, 7):
    print("%-28s +%8.0f cycles  %5.1f %%" % (names[i], st[i] - st[i - 1], 100.0 * (st[i] - st[i - 1]) / tot))
print("total %.0f cycles" % tot)
if 0 < st[7] - st[6] < 1e7: print("kernel's end (flag raised)   +%8.0f cycles" % (st[7] - st[6]))
if st[16] > 0 and 0 < st[6] - st[16] < 1e7: print("tail: step stored, sums formed +%.0f, cameras moved +%.0f, IMU parameters +%.0f, wave sums +%.0f" % (st[16] - st[5], st[17] - st[16], st[18] - st[17], st[6] - st[18]))
if st[8] > 0 and 0 < st[12] - st[8] < 1e7 and st[9] >= st[8]:
    print("one-wavefront solve (D <= 32): load %.0f, factor %.0f, L to LDS %.0f, back-substitution %.0f cycles" % (st[9]-st[8], st[10]-st[9], st[11]-st[10], st[12]-st[11]))
elif st[8:14].sum() > 0:
    tiled = st[8] == 0 and st[11] == 0      # (register-tiled factorisation: one phase; the panel form fills all six)
    print("inside the workgroup-wide solve (D > 32), cycles%s:" % ("" if tiled else " summed over the panels"))
    labels = (["", "load + damping + register-tiled factorisation", "", "", "back-substitution: partial sums", "back-substitution: panel solves"] if tiled else
              ["load + damping", "diagonal blocks (one wavefront)", "rows below the panel", "trailing update", "back-substitution: partial sums", "back-substitution: panel solves"])
    for n, c in zip(labels, st[8:14]):
        if n: print("  %-46s %8.0f cycles" % (n, c))
if st[20] > 0 and 0 < st[26] - st[20] < 1e6:
    print("register-tiled factorisation: matrix in registers, damping added at +%.0f cycles from the solve's start" % (st[14] - st[4]))
    print("  column step 20 (every stamp behind a full wait): column image written +%.0f, barrier +%.0f, column read +%.0f, reciprocal / rsqrt +%.0f, tile updated +%.0f, factor column stored +%.0f cycles" % tuple(st[21 + i] - st[20 + i] for i in range(6)))
