"""MacenkoStainExtractor (stainlib/extraction/macenko_stain_extractor.py:5-44) on the HIP engine."""
from __future__ import annotations

import numpy as np

from ..utils.stain_utils import ABCStainExtractor, _UINT8_MSG, _to_device, is_uint8_image, raise_for_status


class MacenkoStainExtractor(ABCStainExtractor):

    @staticmethod
    def get_stain_matrix(I, luminosity_threshold=0.8, angular_percentile=99):
        """M. Macenko et al., 'A method for normalizing histology slides for quantitative analysis'.

        :param I: RGB uint8 image (H, W, 3).
        :return: (2, 3) float64, unit-norm rows, haematoxylin first."""
        assert is_uint8_image(I), _UINT8_MSG
        from .. import engine
        p = engine.make_params(luminosity_threshold=float(luminosity_threshold),
                               angular_percentile=float(angular_percentile))
        M, _, status = engine.macenko_fit(_to_device(I), params=p)
        raise_for_status(int(status[0]))
        return M[0].cpu().numpy()

    @staticmethod
    def get_stain_matrices(tiles, luminosity_threshold=0.8, angular_percentile=99):
        """Batched extension: (N,H,W,3) uint8 device tensor -> ((N,2,3) float64, (N,) int32 status) tensors."""
        from .. import engine
        p = engine.make_params(luminosity_threshold=float(luminosity_threshold),
                               angular_percentile=float(angular_percentile))
        M, _, status = engine.macenko_fit(tiles, params=p)
        return M, status
