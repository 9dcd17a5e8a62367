"""CPU oracle for the SALVe BEV-render + verify hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is a CPU restatement (numpy + a small C library,
torch-CPU for the verifier) of the reference's algorithm, used only as the checker by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
Nothing under ``salve_amd/`` imports it, and the product path raises if the HIP library
is missing instead of falling back to anything here.

Pinning: the reference is pure Python, so there is no ``oracle/_ref``
build.  The oracle is pinned by
  * the reference's own known-answer tests for this path, re-expressed on the oracle in
    tests/test_oracle_kats.py (z-order masks, hallucination mask 6x6 K=3, sphere table
    directions, BEVParams/Sim2 transform, prune_to_2d_bbox, interpolation early-outs), and
  * golden vectors produced by importing the reference itself in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz): sphere table, z-order, mask,
    Sim2 transforms and end-to-end ``render_bev_image`` intermediates and outputs.
Third-party arithmetic that is NOT under /root/reference and has no reference test
(scipy/Qhull triangle choice in degenerate configurations, OpenCV's uint8 INTER_LINEAR
resize, torchvision's ResNet definition) is restated from the published algorithms;
cv2 and torchvision are absent from this image, so those two legs are "parity unpinned".
"""
