"""CPU oracle for the DISSC inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / timed CPU baseline.  The product
path (``dissc_amd``) never imports this package and fails loudly when the HIP
library is missing.
"""
