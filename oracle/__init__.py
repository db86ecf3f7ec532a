"""CPU oracle for the camera-ISP hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package.  The product
(``taichi_image_amd``) never does; it fails loudly when the HIP library is absent.
"""
