"""Config 4 of BASELINE.json: the Faster-RCNN-OBB R50-FPN detector (test path) that drives the
operators of this package the way the reference's detector drives its own."""
from .theta_obb_rcnn import TEST_CFG, ThetaOBBRCNN, bench_cfg4, build_detector, demo_inputs

__all__ = ['ThetaOBBRCNN', 'TEST_CFG', 'build_detector', 'demo_inputs', 'bench_cfg4']
