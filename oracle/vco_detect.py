"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (and it can only ever be: see below).

numpy restatement of the dot-detection front-end the reference runs per image before the solver sees anything
(VicalibTask::AddImageMeasurements, /root/reference/src/vicalib-task.cc:264-270, parameters :116-122):

    image_processing_[ii].Process(img->data(), w, h, pitch);        // calibu::ImageProcessing
    conic_finder_[ii].Find(image_processing_[ii]);                  // calibu::ConicFinder
    ... conics[i].center ...                                         // what the calibrator is fed (:296)

with  black_on_white = true, at_threshold = 0.9, at_window_ratio = 30.0, conic_min_area = 4.0, conic_min_density = 0.6,
conic_min_aspect = 0.2.

Calibu (arpg/calibu, pinned by nothing: the reference's CMakeLists.txt:47 takes whatever is installed) is NOT in /root/reference, so
this file restates the PUBLISHED algorithms behind those two calls, not Calibu's source:
  * adaptive threshold against the local mean over a window of half-width w / at_window_ratio taken from an integral image
    (Bradley & Roth 2007): a pixel is "dot" if  I * window_area < at_threshold * window_sum;
  * 4-connected components of the dot pixels; candidates pass  area >= conic_min_area,  area / bbox_area >= conic_min_density,
    min(bw, bh) / max(bw, bh) >= conic_min_aspect  and do not touch the image border;
  * per candidate the dual-conic fit on the image GRADIENT inside the (slightly grown) bounding box, Ouellet & Hebert 2009
    ("Precise ellipse estimation without contour point extraction"): every pixel's gradient g at x defines the line
    l = (g_x, g_y, -g.x) tangent to the ellipse; minimise sum |g|^2 (l^T C* l)^2 over the dual conic C* (C*_33 = 1); the centre
    is (C*_13, C*_23).
Constants Calibu may choose differently (gradient operator, box growth, thresholds on |g|) are marked RECONSTRUCTED.  Target grid
matching (TargetGridDot::FindTarget, :274) is not restated.
"""
import numpy as np

GROW = 2            # RECONSTRUCTED: pixels the bounding box is grown by on each side before the fit
MIN_GRAD2 = 1.0     # RECONSTRUCTED: squared gradient magnitude (grey levels^2 per pixel^2, central differences / 2) below which a pixel is ignored


def integral_image(img):
    """S[y, x] = sum of img[:y, :x] (one row / column of zeros in front), exact in int64."""
    s = np.zeros((img.shape[0] + 1, img.shape[1] + 1), dtype=np.int64)
    s[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.int64), axis=0), axis=1)
    return s


def adaptive_threshold(img, at_threshold=0.9, at_window_ratio=30.0):
    h, w = img.shape
    rad = int(w / at_window_ratio)
    S = integral_image(img)
    y, x = np.mgrid[0:h, 0:w]
    x0 = np.maximum(x - rad, 0); x1 = np.minimum(x + rad + 1, w)
    y0 = np.maximum(y - rad, 0); y1 = np.minimum(y + rad + 1, h)
    area = (x1 - x0) * (y1 - y0)
    tot = S[y1, x1] - S[y0, x1] - S[y1, x0] + S[y0, x0]
    return img.astype(np.float64) * area < at_threshold * tot          # True = dot (black on white)


def gradient(img):
    """Central differences / 2, zero on the border."""
    f = img.astype(np.float64)
    gx = np.zeros_like(f); gy = np.zeros_like(f)
    gx[:, 1:-1] = 0.5 * (f[:, 2:] - f[:, :-2])
    gy[1:-1, :] = 0.5 * (f[2:, :] - f[:-2, :])
    return gx, gy


def label4(fg):
    """4-connected components; every component is named by the smallest linear index y * w + x of its pixels (what a union-find
    with min-roots converges to).  Returns the label image (-1 = background)."""
    h, w = fg.shape
    lab = np.where(fg, np.arange(h * w).reshape(h, w), -1)
    while True:
        new = lab.copy()
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            sh = np.full_like(lab, -1)
            ys = slice(max(dy, 0), h + min(dy, 0)); yd = slice(max(-dy, 0), h + min(-dy, 0))
            xs = slice(max(dx, 0), w + min(dx, 0)); xd = slice(max(-dx, 0), w + min(-dx, 0))
            sh[yd, xd] = lab[ys, xs]
            m = fg & (sh >= 0)
            new[m] = np.minimum(new[m], sh[m])
        if np.array_equal(new, lab):
            return lab
        lab = new


def fit_dual_conic(gx, gy, x0, y0, x1, y1, full=False):
    """Centre of the ellipse whose edge runs through the box [x0, x1) x [y0, y1).  Coordinates relative to the box centre."""
    cx, cy = 0.5 * (x0 + x1 - 1), 0.5 * (y0 + y1 - 1)
    M = np.zeros((5, 5)); rhs = np.zeros(5)
    for y in range(y0, y1):
        for x in range(x0, x1):
            a, b = gx[y, x], gy[y, x]
            w2 = a * a + b * b
            if w2 < MIN_GRAD2:
                continue
            c = -(a * (x - cx) + b * (y - cy))
            K = np.array([a * a, a * b, b * b, a * c, b * c])
            M += w2 * np.outer(K, K); rhs += w2 * K * (-(c * c))
    th = np.linalg.solve(M, rhs)
    if not full:
        return cx + 0.5 * th[3], cy + 0.5 * th[4]
    # the dual conic in image coordinates (translation by the box centre), its inverse = the ellipse x^T C x = 0, unit Frobenius
    # norm and C[0, 0] > 0 (calibu::Conic::C)
    A, B, Cq, D, E = th
    Q = np.array([[A, 0.5 * B, 0.5 * D], [0.5 * B, Cq, 0.5 * E], [0.5 * D, 0.5 * E, 1.0]])
    T = np.array([[1.0, 0.0, cx], [0.0, 1.0, cy], [0.0, 0.0, 1.0]])
    Cm = np.linalg.inv(T @ Q @ T.T)
    Cm = Cm / np.linalg.norm(Cm)
    if Cm[0, 0] < 0:
        Cm = -Cm
    return (cx + 0.5 * th[3], cy + 0.5 * th[4]), Cm


def find_conics(img, at_threshold=0.9, at_window_ratio=30.0, min_area=4.0, min_density=0.6, min_aspect=0.2, black_on_white=True, full=False):
    """Centres (x, y) of the detected dots, ordered by the label (= the smallest pixel index) of their component.
    full: -> (centres [n, 2], conics [n, 3, 3], boxes [n, 4] = x0, y0, x1, y1 inclusive)."""
    if not black_on_white:
        img = 255 - img                       # white dots on black: the same detector on the inverted image
    h, w = img.shape
    fg = adaptive_threshold(img, at_threshold, at_window_ratio)
    lab = label4(fg)
    gx, gy = gradient(img)
    out, conics, boxes = [], [], []
    for L in np.unique(lab[lab >= 0]):
        ys, xs = np.nonzero(lab == L)
        x0, x1, y0, y1 = xs.min(), xs.max() + 1, ys.min(), ys.max() + 1
        area = len(xs); bw, bh = x1 - x0, y1 - y0
        if area < min_area or area < min_density * bw * bh or min(bw, bh) < min_aspect * max(bw, bh):
            continue
        if x0 - GROW < 1 or y0 - GROW < 1 or x1 + GROW > w - 1 or y1 + GROW > h - 1:
            continue
        if full:
            c, Cm = fit_dual_conic(gx, gy, x0 - GROW, y0 - GROW, x1 + GROW, y1 + GROW, full=True)
            out.append(c); conics.append(Cm); boxes.append((x0, y0, x1 - 1, y1 - 1))
        else:
            out.append(fit_dual_conic(gx, gy, x0 - GROW, y0 - GROW, x1 + GROW, y1 + GROW))
    if full:
        return np.array(out).reshape(-1, 2), np.array(conics).reshape(-1, 3, 3), np.array(boxes, dtype=np.int64).reshape(-1, 4)
    return np.array(out).reshape(-1, 2)
