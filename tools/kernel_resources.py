"""Per-kernel register / scratch / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Rpass-analysis=kernel-resource-usage x.hip -o /dev/null 2> res.txt; python tools/kernel_resources.py res.txt"""
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    print("%-64s VGPR %4s AGPR %4s scratch %5s occ %2s LDS %6s" % (name[:64], g('VGPRs'), g('AGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'), g('LDS Size [bytes/block]')))
