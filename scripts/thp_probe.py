import numpy as np, time, os
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
def anon_huge():
    for l in open("/proc/self/smaps_rollup"):
        if l.startswith("AnonHugePages"): return l.split()[1]
keep = []
for rep in range(6):
    a = np.empty((256, 100000), order="F")
    t = time.perf_counter(); a[:, ::2] = 0; a[:, 1::2] = 0; dt = time.perf_counter() - t  # first touch of every page (single thread)
    print(f"rep {rep}: first touch of 205 MB by one thread {dt*1e3:.1f} ms; AnonHugePages {anon_huge()} kB; live arrays {len(keep)}")
    keep.append(a)
    if rep == 3:
        keep.append(np.ones((513, 100000)))  # more live memory
