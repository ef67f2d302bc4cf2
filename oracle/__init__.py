"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the confrez OBCA hot path.

PARITY UNPINNED: the reference (XuShenLZ/conflict_rez) carries all arithmetic of
this path in un-vendored, un-pinned third-party binaries (CasADi -> IPOPT -> HSL
MA97, `setup.py:10-24`) that are absent from the build image, ships no recorded
inputs and has no tests on the path.  Nothing in here was checked against
outputs of the reference itself; see DESIGN.md "Oracle".

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import or execute anything in this package.  The product path
(`conflict_rez_amd`) never does.
"""
