"""Planar-arm workload as the oracle sees it -- TEST INFRASTRUCTURE.

The seeded generator and the NumPy definition of the constraint family live in
sco_py_amd/workloads.py (shared by bench.py, the tests and this oracle: the product
never imports oracle/, the oracle may import the product's input generator).  This
module only re-exports them under the name the oracle and the golden generators use."""
from sco_py_amd.workloads import (arm_dist, arm_dist_jac, block_groups, default_points, ee_cost, ee_jac, ee_pos,  # noqa: F401
                                  joint_limit_rows, link_points, make_batch, corridor_program, make_point_problem, make_problem,
                                  make_program_problem, make_quadratic_problem,
                                  point_dist, point_dist_jac, quad_rows, quad_rows_jac, smooth_Q, step_params, velocity_rows)
