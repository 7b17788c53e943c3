"""CPU oracle for the SISUA VAE training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product
(``sisua_amd``) never imports this package and fails loudly when its HIP
library is missing.
"""
