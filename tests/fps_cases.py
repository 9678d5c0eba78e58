"""Seeded point clouds + case list shared by the fps golden generator and the fps tests."""
import numpy as np


def make_cloud(kind, n, seed):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, n])))
    if kind == "gauss":
        p = rng.standard_normal((n, 3), dtype=np.float32) * np.float32(0.05)
    elif kind == "sphere":
        p = rng.standard_normal((n, 3), dtype=np.float32)
        p /= np.linalg.norm(p, axis=1, keepdims=True).astype(np.float32)
    elif kind == "lattice":  # many exact distance ties
        g = np.stack(np.meshgrid(*[np.arange(10, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
        p = g[rng.integers(0, 1000, size=n)] * np.float32(0.01)
    elif kind == "same":  # all-identical points: every distance is 0
        p = np.tile(np.array([[0.1, -0.2, 0.3]], dtype=np.float32), (n, 1))
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(p, dtype=np.float32)


def fps_cases():
    """(name, kind, n, sn, seed, mode) with mode 'center' or a start index as str."""
    cases = []
    for kind in ("gauss", "sphere", "lattice"):
        for n in (64, 1000, 5000, 50000):
            for sn in (8, 32, 64, 256):
                if n == 50000 and kind != "gauss" and sn != 32:
                    continue
                modes = ["center"] + ([str(s) for s in (0, 7, n - 1)] if sn in (8, 32) else [])
                for mode in modes:
                    cases.append((f"{kind}_n{n}_s{sn}_{mode}", kind, n, sn, 11, mode))
    for mode in ("center", "0", "7"):
        cases.append((f"same_n100_s8_{mode}", "same", 100, 8, 3, mode))
        cases.append((f"gauss_n16_s40_{mode}", "gauss", 16, 40, 5, mode))  # sn > N
    cases.append(("gauss_n1_s4_center", "gauss", 1, 4, 7, "center"))
    cases.append(("gauss_n13000_s128_center", "gauss", 13000, 128, 9, "center"))
    cases.append(("gauss_n200000_s64_center", "gauss", 200000, 64, 13, "center"))
    return cases
