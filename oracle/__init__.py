"""CPU oracle for the IIF training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``iif_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / timed CPU baseline.

Parity status: PINNED.  Every function here is checked against outputs of the
reference itself (``/root/reference/classification``, imported on CPU in the
build container) through the golden vectors in ``tests/golden/`` — see
``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``.  The mmdet
half of the reference cannot be imported (mmcv/mmdet absent); its restatement
(`oracle.mmdet_iif`) is pinned by (1) the reduction of its formula to the
classification loss on the cases both define and (2) the known-answer CE
vectors of the reference's own ``tests/test_metrics/test_losses.py:8-32``.
"""
from . import iif_oracle, resnet_oracle, mmdet_iif  # noqa: F401
