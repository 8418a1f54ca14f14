"""VahadaneStainExtractor (stainlib/extraction/vahadane_stain_extractor.py:16-43) on the HIP engine.

The reference calls ``spams.trainDL`` with its default ``iter=-1`` (a 1-second wall-clock budget) and a
random initialisation, so it is not reproducible against itself.  The engine minimises the same
objective (K=2, lambda1=regularizer, non-negative codes and atoms, unit-ball atoms) to convergence
with a deterministic start; see DESIGN.md "Vahadane"."""
from __future__ import annotations

from ..utils.stain_utils import ABCStainExtractor, _UINT8_MSG, _to_device, is_uint8_image, raise_for_status


class VahadaneStainExtractor(ABCStainExtractor):

    @staticmethod
    def get_stain_matrix(I, luminosity_threshold=0.8, regularizer=0.1):
        """A. Vahadane et al., 'Structure-Preserving Color Normalization and Sparse Stain Separation
        for Histological Images'.  :return: (2, 3) float64, unit-norm rows, haematoxylin first."""
        assert is_uint8_image(I), _UINT8_MSG
        from .. import engine
        p = engine.make_params(luminosity_threshold=float(luminosity_threshold), dl_lambda=float(regularizer))
        M, _, status, _ = engine.vahadane_fit(_to_device(I), params=p)
        raise_for_status(int(status[0]))
        return M[0].cpu().numpy()
