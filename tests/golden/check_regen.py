"""Does every fixture regenerate bit-identically?  

    python tests/golden/check_regen.py [generator ...]      # default: every make_golden*.py except the 12x50 pair (minutes each)

Runs the generators, then compares every array of every .npz they wrote with the committed one (`git show HEAD:<file>`): the
zip containers differ (time stamps), the arrays must not.  Prints one line per fixture; exit code 1 on any difference.
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
        print("%-28s %4d arrays, %d differ %s" % (os.path.basename(f), len(new.keys()), len(diff), diff[:4]))
        bad += len(diff)
    sys.exit(1 if bad else 0)
