"""oracle/ -- CPU restatements of the reference's MSDeformAttn path. TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Nothing
under uninext_amd/ imports this package; the product path fails loudly without its HIP library.
Parity pinned against reference-generated fixtures (tests/golden/, tests/test_oracle_golden.py).
"""
