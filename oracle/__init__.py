"""CPU oracle for the CerberusDet hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (torch fp32 / numpy), the arithmetic of the
reference's hot path: model graph, Detect decode, TAL +
DFL/CIoU/BCE loss, NMS, cross-task NMS and the optimizer-step semantics.

Rules (see the task contract):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * the product (``cerberusdet_amd``) never imports it and never falls back
    to it -- it fails loudly when the HIP library is missing;
  * every function cites the reference file:line it follows.

Pinning: the restatement is checked against golden vectors produced by the
real reference imported in the build container (``tools/make_golden.py`` ->
``tests/golden/*.npz|json``), see ``tests/test_oracle_golden.py``.
The one third-party kernel on the path, ``torchvision.ops.nms`` (torchvision
0.20.1, not vendored, not installed), is restated from its documented
semantics in ``oracle/nms.py``; that boundary is "parity unpinned".
"""
