"""curla_amd: MI355X-native CURL + SAC learner (drop-in for the learner path of
paulvantieghem/curla: CurlSacAgent.update & friends, CNNEncoder/PixelEncoder,
ReplayBuffer, random_crop)."""
from .augmentations import ColorJiggle, IdentityAugmentation, NoisyCover, RandomCrop, make_augmentor  # noqa: F401
from .curl_sac import Actor, Critic, CURL, CurlSacAgent, QFunction  # noqa: F401
from .encoder import CNNEncoder, PixelEncoder  # noqa: F401
from .utils import (FrameStack, ReplayBuffer, eval_mode, make_dir, module_hash, preprocess_obs,  # noqa: F401
                    set_seed_everywhere, soft_update_params)

__all__ = ["CurlSacAgent", "Actor", "Critic", "QFunction", "CURL", "CNNEncoder", "PixelEncoder", "ReplayBuffer",
           "RandomCrop", "IdentityAugmentation", "ColorJiggle", "NoisyCover", "make_augmentor", "eval_mode", "set_seed_everywhere",
           "soft_update_params", "FrameStack", "make_dir", "module_hash", "preprocess_obs"]
