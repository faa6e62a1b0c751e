"""stereoformer_amd -- MI355X (gfx950) native stereo cost-volume + disparity-regression path.

Drop-in replacements for the hot-path leaf modules of Magicboomliu/StereoFormer
(``models/utils``, ``models/CrossAttentionCostVolume``, ``models/LocalCostVolume/Attempts``,
``models/IterativeLocalCostVolume/{corr,utils}``): same names, constructor/forward
arguments and output layouts; the bodies are hand-written HIP kernels reached through
the C ABI in ``include/stereo_hip.h`` (``libstereo_hip.so``).  No CPU fallback.
"""
from . import _native
from . import hbm
from . import ops
from .cost_volume import (CostVolume, CostVolumePyramid, build_concat_volume, build_gwc_volume,
                          build_gwc_volume_with_disparity, corr_volume_with_disparity,
                          groupwise_correlation)
from .estimation import (DisparityEstimation, DisparityEstimationWithProb, corr_soft_argmin,
                         disparity_regression, group_mean_soft_argmin)
from .local_cost_volume import (LocalCostVolume, Variance_Based_Local_Cost_Volume, bounds_lookup,
                                build_cost_volume_from_volume, uncertainty_volume)
from .corr import (AlternateCorrBlock, CorrBlock1D, InputPadder, PytorchAlternateCorrBlock1D, bilinear_sampler,
                   coords_grid)
from .cross_attention import CrossAttentionCostVolume
from . import data
from .data import StereoDataset
from .disparity_warper import disp_warp
from .extractor_epilogue import InstanceNormReLU, instance_norm_relu
from .metrics import D1_metric, EPE_Loss, P1_metric, scale_disp, sequence_loss, thres_metric
from .pfm import read_disp, read_pfm, write_pfm
from .pipelines import LowCNNGRULoop, RAFTStereoLoop
from .update_dlc import (BasicGuidanceNet, DisparityUpdateDLC, DisparityUpdateDLCWithMask,
                         DisparityUpdateDLCWithMaskV2, OffsetPredictionHead, gru_blend, gru_reset)
from .update_raft import BasicMotionEncoder_LZH, BasicMultiUpdateBlock_LZH, FlowHead, SepConvGRU, pool2x, pool4x
from .upsample import (ConvAffinityUpsample, interp, resize_bilinear, upflow8, upsample_convex, upsample_convex8,
                       upsample_flow, upsample_simple8)

__all__ = [
    "InstanceNormReLU", "instance_norm_relu", "StereoDataset", "data",
    "D1_metric", "EPE_Loss", "P1_metric", "scale_disp", "sequence_loss", "thres_metric", "read_disp", "read_pfm",
    "write_pfm",
    "LowCNNGRULoop", "RAFTStereoLoop",
    "BasicGuidanceNet", "DisparityUpdateDLC", "DisparityUpdateDLCWithMask", "DisparityUpdateDLCWithMaskV2",
    "OffsetPredictionHead", "gru_blend", "gru_reset", "BasicMotionEncoder_LZH", "BasicMultiUpdateBlock_LZH",
    "FlowHead", "SepConvGRU", "pool2x", "pool4x",
    "AlternateCorrBlock", "CorrBlock1D", "InputPadder", "PytorchAlternateCorrBlock1D", "bilinear_sampler",
    "coords_grid", "CrossAttentionCostVolume", "disp_warp", "ConvAffinityUpsample", "interp", "resize_bilinear",
    "upflow8", "upsample_convex", "upsample_convex8", "upsample_flow", "upsample_simple8",
    "CostVolume", "CostVolumePyramid", "build_concat_volume", "build_gwc_volume", "build_gwc_volume_with_disparity", "corr_volume_with_disparity", "groupwise_correlation",
    "DisparityEstimation", "DisparityEstimationWithProb", "corr_soft_argmin", "disparity_regression",
    "group_mean_soft_argmin", "LocalCostVolume", "Variance_Based_Local_Cost_Volume", "bounds_lookup",
    "build_cost_volume_from_volume", "uncertainty_volume",
]
