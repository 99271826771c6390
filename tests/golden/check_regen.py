"""Does every fixture regenerate bit-identically?  

    python tests/golden/check_regen.py [generator ...]      # default: every make_golden*.py except the 12x50 pair (minutes each)

Runs the generators, then compares every array of every .npz they wrote with the committed one (`git show HEAD:<file>`): the
arrays must not differ.  Prints one line per fixture; exit code 1 on any difference.

One exception, reported separately: the VECTORISED merit values of a `*_merit_log` (one violation sum per constraint group).
The reference adds up a group's expressions while iterating a Python set of objects (prob.py:559-569), whose
order changes from process to process; that is the reference's own arithmetic and no stand-in can pin it, so those values
carry one rounding of freedom (seen: 1 .. 2 ulp).  They are accepted within 4 ulp and counted in the "ulp" column; everything
else -- every assembled QP, every QP solution, status, iteration count, trajectory, the scalar merit values -- is bit-exact
since r04 (the OSQP stand-in solves in the canonical column order, make_golden.py CANON_NX).
"""
import glob, io, os, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

if __name__ == "__main__":
    gens = sys.argv[1:] or sorted(g for g in glob.glob(os.path.join(HERE, "make_golden*.py")) + [os.path.join(HERE, "make_adjudicate.py")]
                                  if "12x50" not in g)
    before = {f: os.path.getmtime(f) for f in glob.glob(os.path.join(HERE, "*.npz"))}
    for g in gens:
        subprocess.check_call([sys.executable, g], cwd=ROOT, stdout=subprocess.DEVNULL)
    bad = 0
    for f in sorted(before):
        if os.path.getmtime(f) == before[f]:
            continue
        rel = os.path.relpath(f, ROOT)
        old = np.load(io.BytesIO(subprocess.check_output(["git", "show", "HEAD:" + rel], cwd=ROOT)), allow_pickle=False)
        new = np.load(f, allow_pickle=False)
        diff = [k for k in new.keys() if k not in old.keys() or old[k].shape != new[k].shape or old[k].tobytes() != new[k].tobytes()]
        diff += [k for k in old.keys() if k not in new.keys()]
        ulp = 0
        for k in list(diff):
            if k.endswith("merit_log") and k in old.keys() and old[k].shape == new[k].shape:
                a, b = old[k], new[k]
                rows = np.flatnonzero((a != b).any(axis=1))
                if (np.array_equal(a[:, :3], b[:, :3]) and (a[rows, 1] == 1.0).all()
                        and (np.abs(a[rows, 3] - b[rows, 3]) <= 4 * np.spacing(np.abs(a[rows, 3]))).all()):
                    diff.remove(k); ulp += len(rows)
        print("%-28s %4d arrays, %d differ %s%s" % (os.path.basename(f), len(new.keys()), len(diff), diff[:4],
                                                    "   (%d vectorised merit values within 4 ulp)" % ulp if ulp else ""))
        bad += len(diff)
    sys.exit(1 if bad else 0)
