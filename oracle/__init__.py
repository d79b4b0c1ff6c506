"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / CPU yardstick.
The product path (``clipcap_amd``) never imports this package and fails loudly if the HIP
extension is missing.
"""
