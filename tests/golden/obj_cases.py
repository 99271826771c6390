"""Cases of tests/golden/make_golden_obj.py: (name, make_problem arguments, Solver attributes)."""
SMALL = dict(d=3, T=6, K=2, O=2)
CASES = [
    ("o0", dict(i=0, ee_cost_weight=0.5, **SMALL), None),
    ("o1", dict(i=1, ee_cost_weight=2.0, **SMALL), None),
    ("o2", dict(i=2, ee_cost_weight=0.5, **SMALL), dict(max_merit_coeff_increases=3, initial_penalty_coeff=10.0)),
    ("o3", dict(i=3, ee_cost_weight=5.0, **SMALL), dict(max_merit_coeff_increases=2, initial_penalty_coeff=1.0)),
    ("o4", dict(i=4, ee_cost_weight=1.0, reach=True, **SMALL), dict(max_merit_coeff_increases=2, initial_penalty_coeff=10.0)),
]
