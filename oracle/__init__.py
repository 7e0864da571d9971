"""CPU oracle for the GPry GP + NORA hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gpry_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / reported baseline.
"""
