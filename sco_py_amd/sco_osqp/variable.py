"""SCO Variable: an array of QP atoms plus its current and saved value.

Mirror of ``sco_py.sco_osqp.variable`` (/root/reference/sco_py/sco_osqp/variable.py).
"""
import numpy as np


class Variable(object):
    """Ordered block of low-level QP variables (variable.py:4-73).

    ``_osqp_vars``   object array of OSQPVar (any shape, usually (n, 1))
    ``_value``       float array of the same shape, or None
    ``_saved_value`` snapshot used for trust regions and roll-back
    """

    def __init__(self, osqp_vars, value=None):
        assert isinstance(osqp_vars, np.ndarray)
        assert len(osqp_vars) > 0
        self._osqp_vars = osqp_vars.copy()          # copy-in (variable.py:19)
        self._value = None
        if value is not None:
            assert osqp_vars.shape == value.shape
            assert isinstance(value, np.ndarray)
            self._value = value.copy()              # copy-in (variable.py:23)
        self._saved_value = None

    def get_osqp_vars(self):
        # the internal array itself, not a copy (variable.py:28-29)
        return self._osqp_vars

    def get_value(self):
        return None if self._value is None else self._value.copy()

    def add_trust_region(self, trust_box_size):
        """Box of half-width ``trust_box_size`` around the SAVED value, written
        into the atoms' bounds (variable.py:37-45)."""
        assert self._saved_value is not None
        lo = self._saved_value - trust_box_size
        hi = self._saved_value + trust_box_size
        for pos, atom in np.ndenumerate(self._osqp_vars):
            atom.set_lower_bound(lo[pos])
            atom.set_upper_bound(hi[pos])

    def update(self):
        """Pull the atoms' solver values into ``_value`` (variable.py:47-60)."""
        fresh = np.zeros(self._osqp_vars.shape)
        for pos, atom in np.ndenumerate(self._osqp_vars):
            if atom.val is None:
                raise ValueError(
                    f"The variable {atom.var_name} does not have a legitimate value"
                )
            fresh[pos] = atom.val
        self._value = fresh

    def save(self):
        assert not np.any(np.isnan(self._value))
        self._saved_value = self._value.copy()

    def restore(self):
        self._value = self._saved_value.copy()
