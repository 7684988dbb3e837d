#!/usr/bin/env python3
"""Deterministic generator of the synthetic stress scene: a tessellated height field
(y = 0.25 sin(3x) cos(2z), nx x nz quads = 2 nx nz triangles, never zero-area), 64 reflective
spheres on an 8 x 8 lattice (every 4th under a scaling transform, i.e. an ellipsoid),
12 point lights on a ring + 4 directional lights, ambient 0.1.  No random numbers.

  python tools/make_synthetic.py OUTDIR [--nx 1000 --nz 500]      # 1,000,000 triangles
writes OUTDIR/synthetic.obj and OUTDIR/synthetic.scene in the reference's text formats.
"""
import argparse
import math
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--nx", type=int, default=1000)
    ap.add_argument("--nz", type=int, default=500)
    ap.add_argument("--spheres", type=int, default=8, help="lattice side (8 -> 64 spheres)")
    ap.add_argument("--name", default="synthetic")
    a = ap.parse_args()
    os.makedirs(a.outdir, exist_ok=True)
    x0, x1, z0, z1 = -6.0, 6.0, -7.0, 3.0
    with open(os.path.join(a.outdir, a.name + ".obj"), "w") as f:
        f.write("# synthetic height field\n")
        for j in range(a.nz + 1):
            z = z0 + (z1 - z0) * j / a.nz
            for i in range(a.nx + 1):
                x = x0 + (x1 - x0) * i / a.nx
                y = 0.25 * math.sin(3.0 * x) * math.cos(2.0 * z)
                f.write(f"v {x:.6f} {y:.6f} {z:.6f}\n")
        w = a.nx + 1
        for j in range(a.nz):
            for i in range(a.nx):
                v00 = j * w + i + 1
                v10, v01, v11 = v00 + 1, v00 + w, v00 + w + 1
                f.write(f"f {v00} {v01} {v11}\n")      # winding: normal (v2-v1)x(v3-v1) points up (+y)
                f.write(f"f {v00} {v11} {v10}\n")
    eye = (0.0, 4.0, 9.0)
    target = (0.0, 0.0, -2.0)
    fwd = [t - e for t, e in zip(target, eye)]
    n = math.sqrt(sum(c * c for c in fwd)); fwd = [c / n for c in fwd]
    right = [fwd[2] * 0 - fwd[1] * 0, 0, 0]
    right = [-fwd[2], 0.0, fwd[0]]                       # up x fwd with up = (0,1,0), then flipped to point right
    n = math.sqrt(sum(c * c for c in right)); right = [-c / n for c in right]
    up = [right[1] * fwd[2] - right[2] * fwd[1], right[2] * fwd[0] - right[0] * fwd[2], right[0] * fwd[1] - right[1] * fwd[0]]
    up = [-c for c in up]
    d, hw, hh = 4.0, 3.2, 1.8
    centre = [e + d * c for e, c in zip(eye, fwd)]

    def corner(sx, sy):
        return " ".join(f"{centre[k] + sx * hw * right[k] + sy * hh * up[k]:.6f}" for k in range(3))

    with open(os.path.join(a.outdir, a.name + ".scene"), "w") as f:
        f.write(f"cam {eye[0]} {eye[1]} {eye[2]}  {corner(-1, -1)}  {corner(1, -1)}  {corner(-1, 1)}  {corner(1, 1)}\n")
        f.write("lta 0.1 0.1 0.1\n")
        for k in range(12):
            ang = 2 * math.pi * k / 12
            f.write(f"ltp {7 * math.cos(ang):.4f} 6 {-2 + 7 * math.sin(ang):.4f}  0.15 0.15 0.15  0\n")
        for dx, dz in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
            f.write(f"ltd {0.4 * dx} -1 {0.4 * dz}  0.1 0.1 0.1\n")
        f.write("mat 0.1 0.1 0.1  0.5 0.6 0.4  0.3 0.3 0.3  20  0.2 0.2 0.2\n")
        f.write(f"obj {a.name}.obj\n")
        k = 0
        for j in range(a.spheres):
            for i in range(a.spheres):
                x = -4.9 + 9.8 * i / max(1, a.spheres - 1)
                z = -6.0 + 8.0 * j / max(1, a.spheres - 1)
                col = f"{0.3 + 0.5 * (i % 3) / 2:.2f} {0.3 + 0.5 * (j % 3) / 2:.2f} {0.8 - 0.4 * ((i + j) % 2):.2f}"
                f.write(f"mat 0.1 0.1 0.1  {col}  0.8 0.8 0.8  60  0.8 0.8 0.8\n")
                if k % 4 == 3:
                    f.write(f"xft {x:.4f} 0.9 {z:.4f}\nxfs 1 0.6 1.4\nsph 0 0 0 0.3\nxfz\n")
                else:
                    f.write(f"sph {x:.4f} 0.9 {z:.4f} 0.3\n")
                k += 1
    print(f"wrote {a.name}.obj ({2 * a.nx * a.nz} triangles) and {a.name}.scene to {a.outdir}")


if __name__ == "__main__":
    main()
