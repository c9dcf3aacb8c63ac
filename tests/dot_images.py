"""TEST INFRASTRUCTURE: deterministic synthetic images of a dot grid (black dots on white, as the reference's targets) seen through a
pinhole camera, with the centre of every dot's IMAGE ELLIPSE known in closed form -- what a conic detector should return.  (The
centre of the projected ellipse is not the projection of the circle's centre under perspective; the conic algebra below gives
the former.)"""
import numpy as np


def render(width=640, height=480, nx=9, ny=6, spacing=0.03, r_large=0.0094, r_small=0.0063, fu=420.0, fv=420.0, seed=0,
           tilt=(0.25, -0.2, 0.1), dist=0.42, ss=8, white=230, black=25, with_conics=False, pattern=None, missing=0.0, with_grid=False):
    """Returns (image u8 [h, w], centres [n, 2] in pixel coordinates (x, y), pixel centres at integers); with_conics: also the
    image ellipses as 3 x 3 matrices (unit Frobenius norm, first entry positive)."""
    rng = np.random.default_rng(seed)
    from scipy.spatial.transform import Rotation as R
    Rcw = R.from_euler("xyz", tilt).as_matrix()
    t = np.array([-0.5 * (nx - 1) * spacing, -0.5 * (ny - 1) * spacing, 0.0])
    t = Rcw @ t + np.array([0.0, 0.0, dist])
    K = np.array([[fu, 0, 0.5 * (width - 1)], [0, fv, 0.5 * (height - 1)], [0, 0, 1.0]])
    H = K @ np.column_stack([Rcw[:, 0], Rcw[:, 1], t])          # plane (X, Y, 1) -> pixels
    Hi = np.linalg.inv(H)
    img = np.full((height, width), float(white))
    centres, conics = [], []
    big = rng.random((ny, nx)) < 0.4 if pattern is None else np.asarray(pattern).astype(bool)
    gone = rng.random((ny, nx)) < missing          # dots that are not drawn (occluded / undetected)
    grid_idx = []
    for j in range(ny):
        for i in range(nx):
            if gone[j, i]:
                continue
            grid_idx.append(j * nx + i)
            X, Y, r = i * spacing, j * spacing, (r_large if big[j, i] else r_small)
            Cc = np.array([[1, 0, -X], [0, 1, -Y], [-X, -Y, X * X + Y * Y - r * r]], dtype=float)     # circle as a conic in the plane
            Ci = Hi.T @ Cc @ Hi                                                                   # its image
            c = -np.linalg.solve(Ci[:2, :2], Ci[:2, 2])
            centres.append(c)
            Cn = Ci / np.linalg.norm(Ci)
            conics.append(Cn if Cn[0, 0] > 0 else -Cn)
            # coverage by supersampling inside a box around the ellipse
            p = H @ np.array([X, Y, 1.0]); p = p[:2] / p[2]
            rad_px = 1.6 * r * max(fu, fv) / dist + 2
            x0, x1 = int(max(p[0] - rad_px, 0)), int(min(p[0] + rad_px + 1, width))
            y0, y1 = int(max(p[1] - rad_px, 0)), int(min(p[1] + rad_px + 1, height))
            sub = (np.arange(ss) + 0.5) / ss - 0.5
            ys, xs = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
            cov = np.zeros(ys.shape)
            for dy in sub:
                for dx in sub:
                    px = np.stack([xs + dx, ys + dy, np.ones_like(xs, dtype=float)], axis=-1)
                    q = np.einsum("...i,ij,...j->...", px, Ci, px)
                    cov += (q * np.sign(Cc[0, 0] * 1.0) < 0) if Ci[0, 0] > 0 else (q > 0)
            cov /= ss * ss
            img[y0:y1, x0:x1] = np.minimum(img[y0:y1, x0:x1], white - (white - black) * cov)
    out = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if with_grid:      # also the target-dot index (row * nx + col) of every dot drawn and the large / small pattern used
        return out, np.array(centres), np.array(conics), np.array(grid_idx), big.astype(np.int32)
    if with_conics:
        return out, np.array(centres), np.array(conics)
    return out, np.array(centres)
