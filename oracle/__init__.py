"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the LAUDNet dynamic-inference hot path (dense emulation in
torch fp32 + numpy integer index work).  It is the checker the HIP path is
compared against; it is never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs may import it.
``laudnet_amd`` must never import from here.

Parity pinning: every function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the
reference's own Python (``/root/reference/imagenet_classification/models``) in
the build container -- see ``tests/golden/make_golden.py``.
"""
