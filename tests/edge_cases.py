import numpy as np
def P(a): return np.array(a, np.int32).reshape(-1, 1, 2)
def cases08(W, H):
    rng = np.random.default_rng(9)
    line = P([[100, 100], [400, 130], [700, 90], [900, 300]])
    c = {}
    c["empty"] = []
    c["single_two_point"] = [P([[10, 10], [300, 200]])]
    c["zero_length"] = [P([[50, 50], [50, 50], [50, 50]]), P([[60, 60], [60, 60]]), line]
    c["exact_duplicates"] = [line, line.copy(), line[::-1].copy(), line.copy()]
    c["only_tiny"] = [P([[20 + 9 * i, 30 + 7 * i], [22 + 9 * i, 31 + 7 * i], [21 + 9 * i, 33 + 7 * i]]) for i in range(40)]
    c["outside_canvas"] = [P([[-500, -500], [-100, -300], [-50, -900]]), P([[W + 50, 10], [W + 400, 300]]), P([[10, H + 20], [500, H + 400]]), line]
    c["crossing_border"] = [P([[-200, 100], [W + 200, 140]]), P([[300, -300], [340, H + 300]]), P([[W - 3, H - 3], [W + 40, H + 40]])]
    c["closed_loops"] = [P([[200, 200], [600, 200], [600, 600], [200, 600], [200, 200]]), P([[203, 201], [598, 203], [601, 597], [199, 602], [203, 201]])]
    c["spiral_self_overlap"] = [P([[int(500 + (5 + t * 0.4) * np.cos(t / 6)), int(500 + (5 + t * 0.4) * np.sin(t / 6))] for t in range(600)])]
    t = np.arange(5000)
    c["long_polyline_5000"] = [P(np.stack([600 + 0.11 * t * np.cos(t / 37.0), 800 + 0.13 * t * np.sin(t / 41.0)], 1).astype(np.int32)), line]
    return c
def cases07():
    rng = np.random.default_rng(3)
    c = {}
    c["single"] = [P([[5, 5], [90, 40], [10, 70]])]
    c["all_equal_two_point"] = [P([[100, 100], [200, 200]]) for _ in range(9)]
    for n in (63, 64, 65, 130):
        c[f"n{n}"] = [P(np.cumsum(rng.integers(-30, 31, (int(rng.integers(2, 9)), 2)), axis=0) + rng.integers(100, 3000, 2)) for _ in range(n)]
    c["shared_endpoints"] = [P([[100 * (i % 7), 100 * (i // 7)], [100 * (i % 7) + 100, 100 * (i // 7)]]) for i in range(70)]
    return c
