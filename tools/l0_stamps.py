"""k_chain_l0 (chain assembly folded into the bottom level), middle group of BASELINE cfg3: when builder 2 hands its frames over and when sweep 0
starts / ends its eliminations (library built with -DVC_L0_STAMPS, VICALIB_AMD_LIB pointing at it).  100 MHz ticks -> microseconds from
the wavefront's first instruction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate_native(synth.BASELINE_CONFIGS["cfg3"])
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
sw = ["own frame e1 built", "e1 taken", "e1 done", "e2 taken", "e2 done", "e3 taken", "e3 done", "mid taken", "mid done", "separator stored (waited)"]
bu = ["e2 handed over", "e3 handed over", "mid handed over", "build loop left"]
for title, off, names in (("sweep 0", 0, sw), ("builder 2", 16, bu)):
    print(title); prev = st[off]
    for i, n in enumerate(names):
        x = st[off + 1 + i]
        if x >= st[off]:
            print("  %-28s %7.2f us  (+%.2f)" % (n, (x - st[off]) / 100.0, (x - prev) / 100.0)); prev = x

nm = ["second frame starts (its loads were requested a frame ago)", "Gram record in LDS, IMU columns written", "next frame's loads issued", "H_pp / g_p", "own block, damping", "columns written", "handed over"]
print("builder 2, second frame"); prev = st[24]
for i, n in enumerate(nm):
    x = st[24 + i]
    if x >= st[16]:
        print("  %-62s %7.2f us  (+%.2f)" % (n, (x - st[16]) / 100.0, (x - prev) / 100.0)); prev = x
