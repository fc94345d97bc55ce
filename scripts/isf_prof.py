import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12).astype(np.float64)
a = a[:2048].reshape(256, 8, 12)
H, W = a[:, :4, :], a[:, 4:, :]
def show(name, x, names):
    tot = x.sum(axis=2)
    print(name, "cycles/wave: mean %.0f  min %.0f  max %.0f" % (tot.mean(), tot.min(), tot.max()))
    for i, n in names.items():
        print("   %-28s mean %8.0f  (%.1f%%)   max %8.0f" % (n, x[:, :, i].mean(), 100 * x[:, :, i].sum() / tot.sum(), x[:, :, i].max()))
show("H waves", H, {0: "loads + P1", 1: "epilogue 1", 2: "P2 + update", 3: "stores / post / top", 4: "await empty", 10: "final barrier wait", 11: "tail"})
show("W waves", W, {9: "vt loads issue / top", 5: "await full", 6: "P3 + ratio", 7: "P4", 8: "post / loop", 10: "final barrier wait", 11: "tail (reduction, slab)"})
for c in range(4):
    print("pair", c, "H total %.0f  W total %.0f" % (H[:, c, :10].sum(axis=1).mean(), W[:, c, :10].sum(axis=1).mean()))
