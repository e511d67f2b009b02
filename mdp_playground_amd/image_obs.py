"""Host-side preparation of the polygon templates for image observations
(ImageMultiDiscrete, /root/reference/mdp_playground/spaces/image_multi_discrete.py:129-270).

The reference draws, per observation, a regular polygon with ``state + 3`` vertices
``(int(cx + R cos(2 pi i / k)), int(cy + R sin(2 pi i / k)))`` with Pillow's
``ImageDraw.polygon`` (:186-245).  Pillow's rasteriser is invariant under integer translation
of the vertex list, so the device only needs one bitmap per *vertex-offset pattern*: the
offsets ``int(c + R cos) - c`` depend on the centre coordinate c only through float rounding
(a handful of classes per axis, usually one).  This module enumerates the classes for every
reachable centre, rasterises one template per (state, radius, x-class, y-class) with Pillow
itself — the same third-party rasteriser the reference calls — and self-checks the translation
invariance it relies on against Pillow for the configured image size.
"""
from __future__ import annotations

import numpy as np
import PIL.Image as Image
import PIL.ImageDraw as ImageDraw


def _vertex_offsets(c, R, k, fn):
    """int(c + R*fn(angle_i)) - c for the k vertices (float arithmetic exactly as :188-193)."""
    return tuple(int(c + R * fn((2 * np.pi / k) * i)) - c for i in range(k))


def _edge_clip(c, R, size):
    """How many of the polygon's 2 R + 1 columns (rows) around centre c lie before 0 / past size - 1.  A quantised shift
    ``(v // q) * q`` rounds towards -inf and can leave the draw's own range -mw + 1 .. mw - 1 (:172-181: q = 4 at 100 x 100
    moves the centre by -32 against mw = 30), so the polygon is cut by the picture's edge when Pillow draws it -- and a rotation
    afterwards samples the CUT picture.  A cut centre is its own template class: the template is cut the same way, and the
    renderers' rule "the source pixel outside the picture reads 0" holds on the template alone."""
    return (max(0, R - c), max(0, c + R - (size - 1)))


def radius_range(params):
    R0 = params["circle_radius"]
    if "scale" in params["transforms"]:
        lo, hi = params["scale_range"]
        r_min = int(np.exp(np.log(lo * R0)))
        r_max = int(np.exp(np.log(hi * R0)))
        return min(r_min, int(lo * R0)) - 1, max(r_max, int(hi * R0)) + 1
    return R0, R0


def centre_range(params, R, size):
    """Centre coordinates reachable along one axis for radius R (:172-181)."""
    c0 = int(size / 2)
    if "shift" not in params["transforms"]:
        return [c0]
    m = size / 2 - R
    lo, hi = int(-m + 1), int(m)            # Generator.integers truncates float bounds toward 0
    if lo >= hi:                            # (the reference's Generator.integers(low, high) raises ValueError: low >= high)
        raise ValueError(f"image transform 'shift': a polygon of radius {R} leaves no shift inside {size} pixels")
    q = params["sh_quant"]
    return sorted({c0 + (v // q) * q for v in range(lo, hi)})


def build_templates(S, params, check=True):
    """Returns dict(tpl uint8[S][nR][ncx][ncy][t][t], cls_x int16[S][nR][W], cls_y int16[S][nR][H],
    r_min, r_max, tpl_size, log_min_r, log_max_r)."""
    W, H = params["width"], params["height"]
    R0 = params["circle_radius"]
    r_min, r_max = radius_range(params)
    r_min = max(r_min, 1)
    nR = r_max - r_min + 1
    half = r_max + 1
    t = 2 * half + 1
    cls_x = np.zeros((S, nR, W), np.int16)
    cls_y = np.zeros((S, nR, H), np.int16)
    patterns = {}          # (s, ri) -> (list of x patterns, list of y patterns)
    ncx = ncy = 1
    for s in range(S):
        k = s + 3
        for ri in range(nR):
            R = r_min + ri
            xs, ys = [], []
            for c in centre_range(params, R, W):
                if not (0 <= c < W):
                    continue
                p = _vertex_offsets(c, R, k, np.cos) + _edge_clip(c, R, W)
                if p not in xs:
                    xs.append(p)
                cls_x[s, ri, c] = xs.index(p)
            for c in centre_range(params, R, H):
                if not (0 <= c < H):
                    continue
                p = _vertex_offsets(c, R, k, np.sin) + _edge_clip(c, R, H)
                if p not in ys:
                    ys.append(p)
                cls_y[s, ri, c] = ys.index(p)
            patterns[(s, ri)] = (xs or [(0,) * (k + 2)], ys or [(0,) * (k + 2)])
            ncx, ncy = max(ncx, len(xs)), max(ncy, len(ys))
    tpl = np.zeros((S, nR, ncx, ncy, t, t), np.uint8)
    for (s, ri), (xs, ys) in patterns.items():
        for ix, px in enumerate(xs):
            for iy, py in enumerate(ys):
                img = Image.new("L", (t, t))
                ImageDraw.Draw(img).polygon([(half + dx, half + dy) for dx, dy in zip(px[:-2], py[:-2])], fill=255)
                arr = np.array(img)
                R = r_min + ri                   # the picture's edge cuts the polygon BEFORE the rotation samples it (_edge_clip)
                arr[:, :half - R + px[-2]] = 0
                arr[:, half + R + 1 - px[-1]:] = 0
                arr[:half - R + py[-2], :] = 0
                arr[half + R + 1 - py[-1]:, :] = 0
                tpl[s, ri, ix, iy] = arr
    out = dict(tpl=tpl, cls_x=cls_x, cls_y=cls_y, r_min=r_min, r_max=r_max, tpl_size=t,
               n_cls_x=ncx, n_cls_y=ncy, log_min_r=0.0, log_max_r=0.0)
    if "scale" in params["transforms"]:
        lo, hi = params["scale_range"]
        out["log_min_r"] = float(np.log(lo * R0))
        out["log_max_r"] = float(np.log(hi * R0))
    if check:
        _self_check(S, params, out)
    return out


def _self_check(S, params, t):
    """Pillow at the true centre == template translated, on a sample of reachable centres."""
    W, H = params["width"], params["height"]
    half = t["tpl_size"] // 2
    rng = np.random.default_rng(0)
    for s in range(S):
        k = s + 3
        for R in sorted({t["r_min"], params["circle_radius"], t["r_max"]}):
            ri = R - t["r_min"]
            cxs, cys = centre_range(params, R, W), centre_range(params, R, H)
            for _ in range(6):
                cx, cy = int(rng.choice(cxs)), int(rng.choice(cys))
                if not (0 <= cx < W and 0 <= cy < H):
                    continue
                img = Image.new("L", (W, H))
                pts = [(int(cx + R * np.cos((2 * np.pi / k) * i)), int(cy + R * np.sin((2 * np.pi / k) * i)))
                       for i in range(k)]
                ImageDraw.Draw(img).polygon(pts, fill=255)
                ref = np.array(img)
                tp = t["tpl"][s, ri, t["cls_x"][s, ri, cx], t["cls_y"][s, ri, cy]]
                mine = np.zeros((H, W), np.uint8)
                for ty in range(t["tpl_size"]):
                    y = ty - half + cy
                    if 0 <= y < H:
                        x0 = cx - half
                        lo, hi = max(0, -x0), min(t["tpl_size"], W - x0)
                        if lo < hi:
                            mine[y, x0 + lo:x0 + hi] = tp[ty, lo:hi]
                if not np.array_equal(ref, mine):
                    raise RuntimeError("Pillow polygon raster is not translation invariant here "
                                       f"(state {s}, R {R}, centre {(cx, cy)}); cannot use templates")


def disc_template(R):
    """(2R+1)^2 raster (uint8, 1 = covered) of Pillow's ellipse with the integer bounding box
    centre +- R: what ImageContinuous draws for the agent and the target
    (/root/reference/mdp_playground/spaces/image_continuous.py:190-207).  Integer bounding boxes make
    the raster independent of the position, so the device needs this one template."""
    T = 2 * R + 1
    img = Image.new("L", (T, T), 0)
    ImageDraw.Draw(img).ellipse([(0, 0), (2 * R, 2 * R)], fill=255)
    t = (np.array(img) != 0).astype(np.uint8)
    # self-check of the translation invariance relied on
    big = Image.new("L", (4 * T, 4 * T), 0)
    ImageDraw.Draw(big).ellipse([(T + 3, T + 5), (T + 3 + 2 * R, T + 5 + 2 * R)], fill=255)
    b = (np.array(big) != 0).astype(np.uint8)
    if not (np.array_equal(b[T + 5:T + 5 + T, T + 3:T + 3 + T], t) and b.sum() == t.sum()):
        raise RuntimeError("Pillow ellipse raster is not translation invariant here; cannot use a template")
    return t


def grid_line_mask(W, H, grid_shape):
    """uint8 [n_sub * W, H] (indexed [x][y], 1 = white): the grid lines ImageContinuous draws for a
    grid env, made with Pillow's draw.line from the reference's own end points
    (/root/reference/mdp_playground/spaces/image_continuous.py:145-165 — which spaces the horizontal
    lines by the x-count of the grid)."""
    out = []
    for offset in range(0, len(grid_shape), 2):
        img = Image.new("L", (W, H), 0)
        d = ImageDraw.Draw(img)
        for i in range(1, grid_shape[0 + offset] + 1):
            x_ = i * W // grid_shape[0 + offset] - 1
            d.line([(x_, H), (x_, 0)], fill=255)
        for j in range(1, grid_shape[1 + offset]):
            y_ = j * H // grid_shape[0 + offset]
            d.line([(W, y_), (0, y_)], fill=255)
        out.append((np.array(img).T != 0).astype(np.uint8))
    return np.ascontiguousarray(np.concatenate(out, axis=0))
