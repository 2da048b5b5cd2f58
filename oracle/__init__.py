"""CPU oracle for the Mini-SGLang attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product
(``mini-sglang_amd/``) never imports this package and fails loudly when its
HIP library is missing.

The oracle is a restatement (not a copy) of the reference's torch-native path:

* ``oracle.attention_ref``  - KV slot write, per-sequence page-indexed gather +
  ``scaled_dot_product_attention`` for decode and extend
  (reference ``layers/attention_backends/torch_native_backend.py:25-276``,
  ``memory/kvcache.py:69-85``), plus a dense float64 formula used as the
  tolerance anchor.
* ``oracle.memory_ref``     - free-page FIFO, page table, ``prepare_for_prefill``
  / ``prepare_for_decode`` index math (reference ``memory/kvcache.py:38-67``,
  ``memory/page_manager.py:21-54``, ``engine/batch.py:244-343``).
* ``oracle.radix_ref``      - paged radix tree (reference
  ``memory/radix_cache.py:52-70,101-321``).

Pinning: the reference has no tests or golden vectors of its own,
so the oracle is pinned against outputs of the reference itself, imported from
``/root/reference`` on CPU by ``tests/golden/gen_golden.py`` and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks every function
here against those fixtures.
"""
