"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the EDMP guided-sampler hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / timed CPU baseline.  The product
package (``edmp_amd``) never imports this package and fails loudly when its HIP library is missing.
"""
