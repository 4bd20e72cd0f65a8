"""CPU oracle for the UNOPose hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``unopose_amd`` (the product) never does.
"""
