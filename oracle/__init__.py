"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Boosting-NeRV conditional-decoder train path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / the timed CPU baseline -- never on the HIP path.

Parity status (see DESIGN.md "Oracle"):
  * pinned  : every function in ``cpu_ref.py`` except MS-SSIM is checked against
              outputs of the real reference imported from /root/reference on CPU
              (``ref_harness.py`` + ``make_goldens.py`` -> ``tests/golden/*.npz``).
  * UNPINNED: ``msssim_ref.py`` restates third-party ``pytorch_msssim==0.2.1``
              (requirements.txt:11 of the reference), which is neither vendored in
              /root/reference nor installed here; the reference holds no tests or
              golden vectors for it.  PARITY UNPINNED for the MS-SSIM term.
"""
