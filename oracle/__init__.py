"""ORACLE -- test infrastructure only.

CPU restatement of the reference's algorithm for the hot path.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this package; nothing under openpvsg_amd/ does.
  blocks3p.py   third-party blocks (mmcv 1.4.0 / mmdet 2.25.0)  -- parity UNPINNED (see its header)
  heads.py      in-repo head / post-process arithmetic            -- pinned by tests/golden/head_*.npz
  relation.py   relation head + metrics                           -- pinned by tests/golden/rel_*.npz
  make_golden.py  (build container only) imports the reference to write tests/golden/
  c/            plain-C restatement of the two scalar kernels (MSDA core, pair scorer)
"""
