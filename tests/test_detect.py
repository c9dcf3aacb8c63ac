"""Dot detection, first slice of SURVEY 8 row f4 (vicalib-task.cc:264-270, parameters :116-122).  Calibu's source is not in the
reference tree, so the bar is (a) the numpy restatement of the published algorithms (oracle/vco_detect.py) finding the centres of
rendered ellipses whose true centres are known in closed form, and (b) the HIP kernels agreeing with that restatement."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import dot_images          # noqa: E402
import vco_detect          # noqa: E402


def _match(found, truth):
    d = np.linalg.norm(found[:, None, :] - truth[None, :, :], axis=2)
    assert len(found) == len(truth) and len(set(d.argmin(axis=1))) == len(truth)
    return d.min(axis=1)


def _blurred(img, sigma):
    from scipy.ndimage import gaussian_filter
    return np.clip(np.rint(gaussian_filter(img.astype(float), sigma)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("seed,tilt", [(0, (0.25, -0.2, 0.1)), (3, (-0.35, 0.3, -0.4))])
def test_restatement_finds_the_rendered_ellipse_centres(seed, tilt):
    """54 dots of two sizes under perspective: every dot found once; centre error of the dual-conic fit 0.06 px at most on crisp
    8-bit edges (rms 0.025), 0.02 px with the blur of a real lens (sigma 0.9 px)."""
    img, truth = dot_images.render(seed=seed, tilt=tilt)
    e = _match(vco_detect.find_conics(img), truth)
    assert e.max() < 0.08 and np.sqrt((e ** 2).mean()) < 0.035
    e = _match(vco_detect.find_conics(_blurred(img, 0.9)), truth)
    assert e.max() < 0.02


def _axes(Cm):
    """Centre, semi-axes (descending) of the ellipse x^T C x = 0."""
    c = -np.linalg.solve(Cm[:2, :2], Cm[:2, 2])
    k = -(Cm[2, 2] + Cm[:2, 2] @ c)                       # (x - c)^T A (x - c) = k
    ev = np.linalg.eigvalsh(Cm[:2, :2] / k)
    return c, np.sort(1.0 / np.sqrt(ev))[::-1]


def test_restatement_recovers_the_ellipses_themselves():
    """calibu::Conic carries the ellipse as a matrix (TargetGridDot::FindTarget uses it): the restatement's conic, in image
    coordinates, has the rendered ellipse's centre (by construction of the fit) and its semi-axes within 2.5 % on blurred edges."""
    img, truth, conics = dot_images.render(seed=3, tilt=(-0.35, 0.3, -0.4), with_conics=True)
    cen, Cs, boxes = vco_detect.find_conics(_blurred(img, 0.9), full=True)
    d = np.linalg.norm(cen[:, None, :] - truth[None, :, :], axis=2)
    for i, j in enumerate(d.argmin(axis=1)):
        c, ax = _axes(Cs[i]); ct, axt = _axes(conics[j])
        np.testing.assert_allclose(c, cen[i], atol=1e-9)
        np.testing.assert_allclose(ax, axt, rtol=0.025)
        assert abs(np.linalg.norm(Cs[i]) - 1.0) < 1e-12 and Cs[i][0, 0] > 0
        assert boxes[i][0] <= truth[j][0] <= boxes[i][2] and boxes[i][1] <= truth[j][1] <= boxes[i][3]


def test_restatement_rejects_what_is_not_a_dot():
    img, truth = dot_images.render()
    img[200:203, 10:300] = 20            # a thin dark line: fails the aspect test
    img[5:9, 5:9] = 20                   # a blob at the border: rejected (no room for the fit)
    img[400:402, 500:502] = 20           # 4 pixels: passes the area test (>= 4), density 1, aspect 1 -> a (tiny) extra conic
    out = vco_detect.find_conics(img)
    assert len(out) == len(truth) + 1


@pytest.mark.gpu
@pytest.mark.parametrize("seed,tilt,blur", [(0, (0.25, -0.2, 0.1), 0.0), (3, (-0.35, 0.3, -0.4), 0.9), (5, (0.1, 0.45, 1.2), 0.6)])
def test_gpu_detector_matches_the_restatement(seed, tilt, blur):
    """The HIP kernels (vc_detect.hip) against the numpy restatement on the same image: same dots in the same order (components are
    named by their smallest pixel index on both sides), centres equal to 1e-7 px (the sums run in a different order), and within
    the method's accuracy of the true ellipse centres."""
    from vicalib_amd.lib import ConicDetector
    img, truth = dot_images.render(seed=seed, tilt=tilt)
    if blur:
        img = _blurred(img, blur)
    ref = vco_detect.find_conics(img)
    det = ConicDetector(img.shape[1], img.shape[0])
    got = det.find(img)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-7)
    e = _match(got, truth)
    assert e.max() < (0.08 if not blur else 0.02)
    # white dots on black through the same kernels, and a pitch larger than the width
    inv = np.zeros((img.shape[0], img.shape[1] + 24), dtype=np.uint8); inv[:, :img.shape[1]] = 255 - img
    det2 = ConicDetector(img.shape[1], img.shape[0]); det2.set_params(black_on_white=False)
    n = np.zeros(1, dtype=np.int32); out = np.zeros((4096, 2))
    import ctypes as C
    rc = det2.L.vc_detector_find(det2.h, inv.ctypes.data_as(C.c_void_p), int(inv.strides[0]), out.ctypes.data_as(C.c_void_p), 4096, n.ctypes.data_as(C.c_void_p))
    assert rc == 0 and n[0] == len(ref)
    np.testing.assert_allclose(out[:n[0]], ref, rtol=0, atol=1e-7)


@pytest.mark.gpu
def test_gpu_detector_handles_extra_blobs_and_empty_images():
    from vicalib_amd.lib import ConicDetector
    img, truth = dot_images.render()
    img[200:203, 10:300] = 20; img[5:9, 5:9] = 20; img[400:402, 500:502] = 20
    det = ConicDetector(img.shape[1], img.shape[0])
    np.testing.assert_allclose(det.find(img), vco_detect.find_conics(img), rtol=0, atol=1e-7)
    assert len(det.find(np.full_like(img, 200))) == 0                      # nothing to find
    assert len(det.find(np.zeros_like(img))) == 0                           # all dark: the local mean is dark too


@pytest.mark.gpu
def test_gpu_detector_on_ragged_sizes_and_components_across_tiles():
    """Labelling runs per 32 x 32 tile in LDS, pieces are joined across tile edges afterwards: an image whose sides are no multiples of
    the tile, dots on tile edges and corners, components that wander through many tiles (a ring, a staircase, a comb) -- same dots, same
    order, same boxes as the restatement (components named by their smallest pixel index on both sides)."""
    from vicalib_amd.lib import ConicDetector
    base, _ = dot_images.render(seed=2, tilt=(0.2, -0.1, 0.3))
    img = np.ascontiguousarray(base[:457, :601])
    yy, xx = np.mgrid[0:img.shape[0], 0:img.shape[1]]
    img[(xx - 300) ** 2 + (yy - 230) ** 2 < 47 ** 2] = 25                       # large disc: its rim is a ring through ~12 tiles
    for k in range(12):                                                          # staircase crossing tile corners
        img[330 + 6 * k:330 + 6 * k + 7, 60 + 8 * k:60 + 8 * k + 9] = 20
    img[20:24, 40:400] = 20                                                      # comb: a spine through 12 tiles ...
    for k in range(20):
        img[24:60, 44 + 18 * k:47 + 18 * k] = 20                                 # ... with teeth through two rows of tiles
    for (cx, cy) in ((32.0, 128.0), (64.0, 96.0), (95.5, 159.5), (511.5, 63.5)):  # dots on tile edges / corners
        img[(xx - cx) ** 2 + (yy - cy) ** 2 < 6.5 ** 2] = 25
    cen, Cs, boxes = vco_detect.find_conics(img, full=True)
    det = ConicDetector(img.shape[1], img.shape[0])
    g_cen, g_C, g_box = det.find_conics(img)
    assert len(cen) > 20 and g_cen.shape == cen.shape
    np.testing.assert_allclose(g_cen, cen, rtol=0, atol=1e-7)
    np.testing.assert_array_equal(g_box, boxes)
    # twice on the same detector: no state left behind
    np.testing.assert_array_equal(det.find(img), g_cen)


@pytest.mark.gpu
def test_gpu_detector_with_more_dots_than_travel_in_the_first_copy():
    """The count and the first 512 records come back in one copy, the rest in a second one: 30 x 24 = 720 small dots."""
    from vicalib_amd.lib import ConicDetector
    w, h = 960, 768
    img = np.full((h, w), 225, dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for j in range(24):
        for i in range(30):
            cx, cy = 20.3 + 31.0 * i, 18.7 + 31.0 * j
            sl = (slice(int(cy) - 8, int(cy) + 9), slice(int(cx) - 8, int(cx) + 9))
            img[sl] = np.where((xx[sl] - cx) ** 2 + (yy[sl] - cy) ** 2 < 5.2 ** 2, 30, img[sl])
    ref = vco_detect.find_conics(img)
    det = ConicDetector(w, h)
    got = det.find(img)
    assert len(ref) == 720 and got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-7)
    assert len(det.find(img, max_conics=100)) == 100


@pytest.mark.gpu
def test_gpu_detector_returns_the_conics_and_boxes():
    """vc_detector_find_conics: the ellipse matrices and bounding boxes of calibu::Conic, equal to the restatement's (the 3 x 3
    inverse and the normalisation in a different order of operations: 1e-9 on unit-norm matrices)."""
    from vicalib_amd.lib import ConicDetector
    img, truth, conics = dot_images.render(seed=5, tilt=(0.1, 0.45, 1.2), with_conics=True)
    img = _blurred(img, 0.6)
    cen, Cs, boxes = vco_detect.find_conics(img, full=True)
    det = ConicDetector(img.shape[1], img.shape[0])
    g_cen, g_C, g_box = det.find_conics(img)
    np.testing.assert_allclose(g_cen, cen, rtol=0, atol=1e-7)
    np.testing.assert_allclose(g_C, Cs, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(g_box, boxes)
    np.testing.assert_allclose(det.find(img), g_cen, rtol=0, atol=0)
    d = np.linalg.norm(g_cen[:, None, :] - truth[None, :, :], axis=2)
    for i, j in enumerate(d.argmin(axis=1)):
        _, ax = _axes(g_C[i]); _, axt = _axes(conics[j])
        np.testing.assert_allclose(ax, axt, rtol=0.03)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,tilt,blur", [(0, (0.25, -0.2, 0.1), 0.0), (3, (-0.35, 0.3, -0.4), 0.9), (7, (0.4, 0.1, 2.9), 0.6)])
def test_gpu_detector_to_grid_association(seed, tilt, blur):
    """The whole image front-end on one rendered view: GPU dot detector -> vc_target_find.  Every dot of the 13 x 9 target (10 % of them
    not drawn) gets its true index; what FindTarget's `ellipse_target_map` holds in the reference (vicalib-task.cc:274-277)."""
    from vicalib_amd import lib
    pat = lib.target_make_pattern(9, 13, seed=71)
    img, truth, conics, grid_idx, big = dot_images.render(seed=seed, tilt=tilt, dist=0.45, nx=13, ny=9, spacing=0.022, r_large=0.0069, r_small=0.0046,
                                                          pattern=pat, missing=0.10, with_grid=True, ss=4)
    if blur:
        img = _blurred(img, blur)
    det = lib.ConicDetector(img.shape[1], img.shape[0])
    cen, con, box = det.find_conics(img)
    assert len(cen) == len(truth)
    got, m = lib.target_find(cen, con.reshape(-1, 9), pat)
    assert m >= len(truth) - 2
    d = np.linalg.norm(cen[:, None, :] - truth[None, :, :], axis=2)
    nearest = d.argmin(axis=1)
    hit = got >= 0
    np.testing.assert_array_equal(got[hit], grid_idx[nearest][hit])
