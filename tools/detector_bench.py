#!/usr/bin/env python
"""Images per second through vc_detector_find (host image in, dot centres out) on the two sizes bench.py's `secondary.detector` quotes.
    python tools/detector_bench.py [reps]
With rocprofv3 --kernel-trace --stats around it the per-kernel times of the front-end come out (tools/rocpd_stats.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vicalib_amd.lib import ConicDetector      # noqa: E402


def render(w, h, nx, ny, r):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), 230.0)
    for j in range(ny):
        for i in range(nx):
            cx, cy = (i + 1.0) * w / (nx + 1.0), (j + 1.0) * h / (ny + 1.0)
            rr = r if (i * 7 + j * 3) % 3 else 0.66 * r
            x0, x1, y0, y1 = int(cx - rr - 2), int(cx + rr + 3), int(cy - rr - 2), int(cy + rr + 3)
            d2 = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2
            img[y0:y1, x0:x1] = np.where(d2 < rr * rr, 25.0, img[y0:y1, x0:x1])
    return img.astype(np.uint8)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for (w, h, nx, ny, r) in ((640, 480, 13, 9, 9.0), (1280, 960, 26, 18, 9.0)):
        img = render(w, h, nx, ny, r)
        det = ConicDetector(w, h, 0); det.set_params()
        found = det.find(img)
        for _ in range(10):
            det.find(img)
        t0 = time.perf_counter()
        for _ in range(reps):
            det.find(img)
        dt = time.perf_counter() - t0
        print("%d x %d: %d of %d dots, %.1f images/s, %.4f ms per image" % (w, h, len(found), nx * ny, reps / dt, 1e3 * dt / reps))
        det.close()


if __name__ == "__main__":
    main()
