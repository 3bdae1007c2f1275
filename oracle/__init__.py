"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's arithmetic, used as the checker by tests/, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``.  Nothing in the product
package (``multimodal-sentiment-analysis_amd/``) imports this directory.
"""
