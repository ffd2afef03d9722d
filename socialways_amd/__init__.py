"""socialways_amd - the Social Ways GAN training inner loop (crowdbotp/socialways train.py) as
hand-written HIP for MI355X (gfx950) behind the reference's own Python surface.

    from socialways_amd import Generator, Discriminator, predict, get_traj_4d, SocialWaysTrainer
"""
from .model import (AttentionPooling, DecoderFC, Discriminator, EmbedSocialFeatures, EncoderLstm, Generator,  # noqa: F401
                    SocialFeatures, get_traj_4d, predict, predict_cv, set_default_generator)
from ._lib import SocialWaysHipError, load as load_library  # noqa: F401
from .trainer import SocialWaysTrainer  # noqa: F401
from .data import (SceneDataset, Scale, synth_tracks, toy_tracks, shard_scenes, ragged_scene_sizes,  # noqa: F401
                   parse_biwi, create_dataset, biwi_to_npz, write_biwi_obsmat, synth_crowd_frames)
from . import stats  # noqa: F401

__all__ = ["AttentionPooling", "DecoderFC", "Discriminator", "EmbedSocialFeatures", "EncoderLstm", "Generator",
           "SocialFeatures", "get_traj_4d", "predict", "predict_cv", "set_default_generator", "SocialWaysHipError",
           "load_library", "SocialWaysTrainer", "SceneDataset", "Scale", "synth_tracks", "toy_tracks", "shard_scenes", "ragged_scene_sizes", "stats",
           "parse_biwi", "create_dataset", "biwi_to_npz", "write_biwi_obsmat", "synth_crowd_frames"]
