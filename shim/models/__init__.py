"""Import shim: ``models`` as the reference's callers spell it, served by pointdsc_amd.

Put this directory FIRST on the path and the reference's evaluation scripts run on the HIP path with no edit at all:

    PYTHONPATH=/path/to/pointdsc_amd_repo/shim:/path/to/pointdsc_amd_repo:$PYTHONPATH python evaluation/test_3DMatch.py ...

``from models.PointDSC import PointDSC`` (evaluation/test_3DMatch.py:213, test_KITTI.py:181, test_3DLoMatch.py:266,
demo_registration.py:5, multiway/test_multi_ate.py:330, multiway/test_multi.py:185) then yields ``pointdsc_amd.PointDSC``, and
``from models.common import rigid_transform_3d`` (multiway/test_multi_ate.py:17, baseline_scripts/*.py) the device Kabsch solver.
Nothing else of the reference's ``models`` package is provided: the training loop needs autograd (out of scope).
"""
