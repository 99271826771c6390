"""Scheduling model of the time-sliced ADMM launches (no GPU): 256 CUs, one workgroup per CU at a time,
workgroups dispatched in list order, per-problem QP iteration chains from gpurun_out/qp_iters_B.npy
(scripts/gpu_dump_traces.py).  Compares the fixed slice with 'problems that are ahead run longer'."""
import sys, heapq
import numpy as np

US_PER_IT = 1.09e-3      # ms per ADMM iteration of one workgroup (r02: 1.06-1.10 us incl. the termination tests)
ROUND_OVERHEAD = 0.35    # ms of pre, setup, post + host round trip per round (r02 kernel trace: 0.1-0.7)
CUS = 256


def launch_time(durs):
    if len(durs) <= CUS:
        return max(durs) if len(durs) else 0.0
    h = [0.0] * CUS
    heapq.heapify(h)
    for d in durs:
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)


def simulate(chains, base, mult, lpt, cap=16):
    B = len(chains)
    qi = [0] * B          # index of the QP being solved
    left = [c[0] for c in chains]
    active = set(range(B))
    t = 0.0; rounds = 0
    while active:
        kmin = min(qi[b] for b in active)
        items = []
        for b in active:
            s = base * min(cap, mult ** (qi[b] - kmin)) if base else 10 ** 9
            items.append((min(s, left[b]), b))
        if lpt:
            items.sort(key=lambda z: -z[0])
        else:
            items.sort(key=lambda z: z[1])
        t += launch_time([it * US_PER_IT for it, _ in items]) + ROUND_OVERHEAD
        rounds += 1
        for it, b in items:
            left[b] -= it
            if left[b] <= 0:
                qi[b] += 1
                if qi[b] >= len(chains[b]):
                    active.discard(b)
                else:
                    left[b] = chains[b][qi[b]]
    return t, rounds


if __name__ == "__main__":
    f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qp_iters_1024.npy"
    a = np.load(f)
    chains = [[int(v) for v in row[1:] if v > 0] for row in a]     # column 0 = the projection QP (round 0)
    chains = [c for c in chains if c]
    tot = sum(sum(c) for c in chains)
    print("problems", len(chains), "ideal", tot * US_PER_IT / CUS, "longest chain", max(sum(c) for c in chains) * US_PER_IT)
    for base in (0, 6250, 3125, 12500):
        for mult, lpt in ((1, False), (4, False), (4, True), (2, True), (8, True), (16, True)):
            if base == 0 and (mult > 1 or lpt):
                continue
            print("slice", base, "mult", mult, "lpt", lpt, "-> %.0f ms, %d rounds" % simulate(chains, base, mult, lpt))


def simulate_balanced(chains, base, pick="index", kmax=8):
    """Two slice lengths per launch so that every CU gets the same total: with A active problems and
    k = ceil(A/256) passes, x = A - 256 (k-1) CUs run k problems of `base` and the others k-1 problems of
    base k/(k-1); the long ones are dispatched first."""
    B = len(chains)
    qi = [0] * B
    left = [c[0] for c in chains]
    done_in_qp = [0] * B
    active = set(range(B))
    t = 0.0; rounds = 0
    while active:
        A = len(active)
        k = -(-A // CUS)
        ids = sorted(active)
        if k >= 2 and k <= kmax and A % CUS:
            x = A - CUS * (k - 1)
            n_long = (CUS - x) * (k - 1)
            long_s = (base * k // (k - 1)) // 25 * 25
            if pick == "progress":
                ids.sort(key=lambda b: -done_in_qp[b])
            elif pick == "oracle":
                ids.sort(key=lambda b: -(left[b] + sum(chains[b][qi[b] + 1:])))
            items = [(min(long_s if i < n_long else base, left[b]), b) for i, b in enumerate(ids)]
        elif A <= CUS:
            items = [(min(base * 2, left[b]), b) for b in ids]
        else:
            items = [(min(base, left[b]), b) for b in ids]
        t += launch_time([it * US_PER_IT for it, _ in items]) + ROUND_OVERHEAD
        rounds += 1
        for it, b in items:
            left[b] -= it; done_in_qp[b] += it
            if left[b] <= 0:
                qi[b] += 1; done_in_qp[b] = 0
                if qi[b] >= len(chains[b]):
                    active.discard(b)
                else:
                    left[b] = chains[b][qi[b]]
    return t, rounds


if __name__ == "__main__":
    for base in (6300, 3150, 12600):
        for pick in ("index", "progress", "oracle"):
            print("balanced base", base, pick, "-> %.0f ms, %d rounds" % simulate_balanced(chains, base, pick))


def simulate_catchup(chains, base, mult, lpt, cap=16):
    """Problems that are behind in QP count (their QPs take longer) get longer slices."""
    B = len(chains)
    qi = [0] * B
    left = [c[0] for c in chains]
    active = set(range(B))
    t = 0.0; rounds = 0
    while active:
        kmax = max(qi[b] for b in active)
        items = []
        for b in sorted(active):
            s = base * min(cap, mult ** (kmax - qi[b]))
            items.append((min(s, left[b]), b))
        if lpt:
            items.sort(key=lambda z: -z[0])
        t += launch_time([it * US_PER_IT for it, _ in items]) + ROUND_OVERHEAD
        rounds += 1
        for it, b in items:
            left[b] -= it
            if left[b] <= 0:
                qi[b] += 1
                if qi[b] >= len(chains[b]):
                    active.discard(b)
                else:
                    left[b] = chains[b][qi[b]]
    return t, rounds


if __name__ == "__main__":
    for base in (6250, 12500):
        for mult, lpt in ((2, False), (2, True), (3, True), (4, True), (4, False)):
            print("catch-up base", base, "mult", mult, "lpt", lpt, "-> %.0f ms, %d rounds" % simulate_catchup(chains, base, mult, lpt))


def simulate_queue(chains, base, K):
    """Device-side queue inside a launch: every active problem may run up to K slices per launch; a CU that becomes
    free takes the next ready slice (a problem's next slice is ready when its previous one has finished); a problem
    whose QP ends leaves the launch (post / pre / setup happen between launches)."""
    B = len(chains)
    qi = [0] * B
    left = [c[0] for c in chains]
    active = set(range(B))
    t = 0.0; rounds = 0
    while active:
        # event simulation of one launch
        ready = [(0.0, b) for b in sorted(active)]          # (time the problem becomes ready, problem)
        heapq.heapify(ready)
        cus = [0.0] * CUS
        heapq.heapify(cus)
        done_slices = {b: 0 for b in active}
        end = 0.0
        finished_qp = []
        while ready:
            rt, b = heapq.heappop(ready)
            cu = heapq.heappop(cus)
            start = max(rt, cu)
            it = min(base, left[b])
            fin = start + it * US_PER_IT
            heapq.heappush(cus, fin)
            end = max(end, fin)
            left[b] -= it; done_slices[b] += 1
            if left[b] <= 0:
                finished_qp.append(b)
            elif done_slices[b] < K:
                heapq.heappush(ready, (fin, b))
        t += end + ROUND_OVERHEAD
        rounds += 1
        for b in finished_qp:
            qi[b] += 1
            if qi[b] >= len(chains[b]): active.discard(b)
            else: left[b] = chains[b][qi[b]]
    return t, rounds


if __name__ == "__main__":
    for base in (6250, 3125, 1500):
        for K in (1, 2, 4, 8, 16):
            print("queue base", base, "K", K, "-> %.0f ms, %d rounds" % simulate_queue(chains, base, K))
