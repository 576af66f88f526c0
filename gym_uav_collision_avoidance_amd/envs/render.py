"""Software rasteriser for render(mode="rgb_array") (MUW:243-331 draws the same primitives with pygame)."""
import math

import numpy as np

MAX_WINDOW = 800  # MUW:25


def _disc(img, cx, cy, r, color, width=0):
    h, w, _ = img.shape
    x0, x1 = max(0, int(cx - r - 1)), min(w, int(cx + r + 2))
    y0, y1 = max(0, int(cy - r - 1)), min(h, int(cy + r + 2))
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    d2 = (xx - cx) ** 2 + (yy - cy) ** 2
    mask = d2 <= r * r
    if width:
        mask &= d2 >= max(r - width, 0) ** 2
    img[y0:y1, x0:x1][mask] = color


def _line(img, x0, y0, x1, y1, color, width=1):
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs, ys = np.linspace(x0, x1, n), np.linspace(y0, y1, n)
    h, w, _ = img.shape
    for dx in range(-(width // 2), width // 2 + 1):
        for dy in range(-(width // 2), width // 2 + 1):
            xi, yi = np.round(xs + dx).astype(int), np.round(ys + dy).astype(int)
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            img[yi[ok], xi[ok]] = color


def draw_world(loc, tgt, vel, colors, x_size, y_size, collider_radius, d_sense):
    if x_size > y_size:  # MUW:29-34
        wx, wy = MAX_WINDOW, int(MAX_WINDOW / x_size * y_size)
    else:
        wy, wx = MAX_WINDOW, int(MAX_WINDOW / y_size * x_size)
    img = np.full((wy, wx, 3), 255, dtype=np.uint8)
    ppm = wx / x_size
    size = 10  # object_render_size, MUW:254

    def px(p):
        return (p[0] + x_size / 2) * ppm, wy - (p[1] + y_size / 2) * ppm

    n = len(loc)
    for i in range(n):
        tx, ty = px(tgt[i])
        x0, y0 = int(tx - size / 2), int(ty - size / 2)
        img[max(0, y0):max(0, y0 + size), max(0, x0):max(0, x0 + size)] = colors[i]   # target square
        ax, ay = px(loc[i])
        _disc(img, ax, ay, size, colors[i])                                           # UAV
        th = math.atan2(-vel[i][1], vel[i][0])
        _line(img, ax, ay, ax + size * math.cos(th), ay + size * math.sin(th), (0, 0, 0), width=3)
        _disc(img, ax, ay, collider_radius * ppm, colors[i], width=1)                 # collider ring
    a0 = np.asarray(loc[0], dtype=np.float32)
    d = np.array([np.sqrt(np.sum((np.asarray(loc[j], np.float32) - a0) ** 2)) if j else np.inf for j in range(n)])
    near = [j for j in np.argsort(d, kind="stable") if j and d[j] < d_sense][:2]
    ax, ay = px(loc[0])
    for j in near:
        ox, oy = px(loc[j])
        _line(img, ax, ay, ox, oy, (255, 0, 0))
    _disc(img, ax, ay, d_sense * ppm, colors[0], width=1)                             # sensing ring
    return img
