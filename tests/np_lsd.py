"""A second, independent statement of the LSD detector core (cv::LineSegmentDetector with lsd_refine = 0 or 1) for SMALL images, written
from the published algorithm (von Gioi et al., IPOL 2012) in numpy + plain Python loops — the cross-check of oracle/stvo_lsd_oracle.c
that a reference vector would otherwise provide (parity with OpenCV itself is unpinned: DESIGN.md §3).  It shares no code with the C
oracle; where the two must agree to the last bit it follows the same published definitions:
  * level-line angle and region angle: cv::fastAtan2's float polynomial, evaluated in numpy float32 operation by operation;
  * the cos / sin added to a region's sums and the rectangle's direction: the routine the oracle's header defines (Cody-Waite reduction
    by pi / 2 with two FUSED multiply-adds, fdlibm kernel polynomials) — the fused operations are emulated exactly with fractions;
  * pseudo-ordering: bins from the highest, row-major inside a bin.
Scale 1 only (the blur / resize in front of the core are the 8-bit fixed-point forms tests/test_oracle_orb.py checks separately)."""
from fractions import Fraction
import math

import numpy as np

F32 = np.float32
DEG2RAD = math.pi / 180.0
M_3_2_PI = (3 * math.pi) / 2
M_2_PI = 2 * math.pi


def fast_atan2_deg(y, x):
    y, x = F32(y), F32(x)
    scale = F32(180.0 / math.pi)
    p1, p3, p5, p7 = (F32(0.9997878412794807) * scale, F32(-0.3258083974640975) * scale, F32(0.1555786518463281) * scale,
                      F32(-0.04432655554792128) * scale)
    ax, ay = np.abs(x), np.abs(y)
    eps = F32(2.2204460492503131e-16)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = F32(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = F32(180.0) - a
    if y < 0:
        a = F32(360.0) - a
    return F32(a)


def _fma(a, b, c):
    """round(a * b + c) with ONE rounding: exact rational arithmetic, then the nearest double (ties to even)."""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def sincos_det(x):
    PIO2_HI, PIO2_LO, TWO_OVER_PI = 1.57079632679489655800e+00, 6.12323399573676603587e-17, 6.36619772367581382433e-01
    S = (-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
         -2.50507602534068634195e-08, 1.58969099521155010221e-10)
    C = (4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
         2.08757232129817482790e-09, -1.13596475577881948265e-11)
    k = float(np.rint(x * TWO_OVER_PI))
    r = _fma(-k, PIO2_HI, x)
    r = _fma(-k, PIO2_LO, r)
    z = r * r
    ps = S[1] + z * (S[2] + z * (S[3] + z * (S[4] + z * S[5])))
    sn = r + (z * r) * (S[0] + z * ps)
    pc = z * (C[0] + z * (C[1] + z * (C[2] + z * (C[3] + z * (C[4] + z * C[5])))))
    cs = 1.0 - (0.5 * z - z * pc)
    q = int(k) & 3
    return ((sn, cs, -sn, -cs)[q], (cs, -sn, -cs, sn)[q])


def segments(img, quant=2.0, ang_th=22.5, n_bins=1024, refine=0, density_th=0.6, stats=None):
    """[n, 4] float32 (x1, y1, x2, y2) in detection order, for an 8-bit image at scale 1.  refine: 0 (LSD_REFINE_NONE) or 1
    (LSD_REFINE_STD: a sparse region is given back, grown again under a tolerance from the angles near its seed, then cut back by
    radius).  stats (a dict): how often the branches of the refinement ran."""
    def count(key):
        if stats is not None:
            stats[key] = stats.get(key, 0) + 1
    img = np.asarray(img, np.int64)
    h, w = img.shape
    prec = math.pi * ang_th / 180
    p = ang_th / 180
    rho = quant / math.sin(prec)
    # level lines
    DA = img[1:, 1:] - img[:-1, :-1]
    BC = img[:-1, 1:] - img[1:, :-1]
    gx, gy = DA + BC, DA - BC
    norm = np.sqrt((gx * gx + gy * gy) / 4.0)
    mod = np.zeros((h, w))
    mod[:-1, :-1] = norm
    ang = np.full((h, w), -1024.0)
    ys, xs = np.nonzero(norm > rho)
    for y, x in zip(ys, xs):
        ang[y, x] = float(fast_atan2_deg(gx[y, x], -gy[y, x])) * DEG2RAD
    defined = ang != -1024.0
    max_grad = norm[norm > rho].max() if len(ys) else -1.0
    bin_coef = (n_bins - 1) / max_grad if max_grad > 0 else 0.0
    bins = np.clip((mod[:-1, :-1] * bin_coef).astype(np.int64), 0, n_bins - 1)
    order = np.lexsort((np.arange((h - 1) * (w - 1)), -bins.ravel()))  # highest bin first, row-major inside a bin
    oy, ox = np.divmod(order, w - 1)
    log_nt = 5 * (math.log10(w) + math.log10(h)) / 2 + math.log10(11.0)
    min_reg = int(-log_nt / math.log10(p))
    used = np.zeros((h, w), bool)
    out = []

    def aligned(x, y, theta, tol):
        a = ang[y, x]
        if a == -1024.0:
            return False
        d = abs(theta - a)
        if d > M_3_2_PI:
            d = abs(d - M_2_PI)
        return d <= tol

    def grow(sx, sy, tol):
        reg = [(sx, sy)]
        reg_angle = ang[sy, sx]
        sn, cs = sincos_det(reg_angle)
        sumdx, sumdy = F32(cs), F32(sn)
        used[sy, sx] = True
        i = 0
        while i < len(reg):
            px, py = reg[i]
            for yy in range(max(py - 1, 0), min(py + 1, h - 1) + 1):
                for xx in range(max(px - 1, 0), min(px + 1, w - 1) + 1):
                    if not used[yy, xx] and aligned(xx, yy, reg_angle, tol):
                        used[yy, xx] = True
                        reg.append((xx, yy))
                        sn, cs = sincos_det(float(F32(ang[yy, xx])))
                        sumdx = F32(sumdx + F32(cs))
                        sumdy = F32(sumdy + F32(sn))
                        reg_angle = float(fast_atan2_deg(sumdy, sumdx)) * DEG2RAD
            i += 1
        return reg, reg_angle

    def rect(reg, reg_angle):
        x = y = s = 0.0
        for px, py in reg:
            wgt = mod[py, px]
            x += float(px) * wgt
            y += float(py) * wgt
            s += wgt
        x /= s
        y /= s
        Ixx = Iyy = Ixy = 0.0
        for px, py in reg:
            wgt = mod[py, px]
            ddx, ddy = float(px) - x, float(py) - y
            Ixx += ddy * ddy * wgt
            Iyy += ddx * ddx * wgt
            Ixy -= ddx * ddy * wgt
        lam = 0.5 * (Ixx + Iyy - math.sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy))
        theta = float(fast_atan2_deg(F32(lam - Ixx), F32(Ixy))) if abs(Ixx) > abs(Iyy) else float(fast_atan2_deg(F32(Ixy), F32(lam - Iyy)))
        theta *= DEG2RAD
        diff = theta - reg_angle
        while diff <= -math.pi:
            diff += M_2_PI
        while diff > math.pi:
            diff -= M_2_PI
        if abs(diff) > prec:
            theta += math.pi
        dy, dx = sincos_det(theta)
        l_min = l_max = w_min = w_max = 0.0
        for px, py in reg:
            l = (float(px) - x) * dx + (float(py) - y) * dy
            ww = -(float(px) - x) * dy + (float(py) - y) * dx
            l_max = max(l_max, l)
            l_min = min(l_min, l)
            w_max = max(w_max, ww)
            w_min = min(w_min, ww)
        return [x + l_min * dx, y + l_min * dy, x + l_max * dx, y + l_max * dy, max(w_max - w_min, 1.0)]

    def dist_sq(x1, y1, x2, y2):
        return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)

    def density(rc, n):
        return float(n) / (math.sqrt(dist_sq(rc[0], rc[1], rc[2], rc[3])) * rc[4])

    def refine_region(reg, reg_angle, rc):
        """-> (reg, reg_angle, rc) or None when the region is given up"""
        if density(rc, len(reg)) >= density_th:
            return reg, reg_angle, rc
        count("regrown")
        xc, yc = float(reg[0][0]), float(reg[0][1])
        ang_c = ang[reg[0][1], reg[0][0]]
        tot = sq = 0.0
        n = 0
        for px, py in reg:
            used[py, px] = False
            if math.sqrt(dist_sq(xc, yc, float(px), float(py))) < rc[4]:
                d = ang[py, px] - ang_c
                while d <= -math.pi:
                    d += M_2_PI
                while d > math.pi:
                    d -= M_2_PI
                tot += d
                sq += d * d
                n += 1
        mean = tot / float(n)
        arg = (sq - 2.0 * mean * tot) / float(n) + mean * mean
        tau = 2.0 * math.sqrt(arg) if arg >= 0 else float("nan")
        reg, reg_angle = grow(reg[0][0], reg[0][1], tau)
        if len(reg) < 2:
            count("gone_after_regrowing")
            return None
        rc = rect(reg, reg_angle)
        dens = density(rc, len(reg))
        if dens >= density_th:
            return reg, reg_angle, rc
        count("radius_reduced")
        rad_sq = max(dist_sq(xc, yc, rc[0], rc[1]), dist_sq(xc, yc, rc[2], rc[3]))
        while dens < density_th:
            rad_sq *= 0.75 * 0.75
            count("radius_steps")
            i = 0
            while i < len(reg):
                if dist_sq(xc, yc, float(reg[i][0]), float(reg[i][1])) > rad_sq:
                    used[reg[i][1], reg[i][0]] = False
                    reg[i] = reg[-1]
                    reg.pop()
                else:
                    i += 1
            if len(reg) < 2:
                count("gone_by_radius")
                return None
            rc = rect(reg, reg_angle)
            dens = density(rc, len(reg))
        return reg, reg_angle, rc

    for sy, sx in zip(oy, ox):
        if used[sy, sx] or not defined[sy, sx]:
            continue
        reg, reg_angle = grow(sx, sy, prec)
        if len(reg) < min_reg:
            continue
        rc = rect(reg, reg_angle)
        if refine >= 1:
            res = refine_region(reg, reg_angle, rc)
            if res is None:
                continue
            reg, reg_angle, rc = res
        out.append([F32(rc[0] + 0.5), F32(rc[1] + 0.5), F32(rc[2] + 0.5), F32(rc[3] + 0.5)])
    return np.array(out, np.float32).reshape(-1, 4)
