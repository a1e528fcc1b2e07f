"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's CQ / KZG proving path.

Import rules: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package; the product never does.
"""
