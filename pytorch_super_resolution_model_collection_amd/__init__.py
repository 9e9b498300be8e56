"""MI355X-native convolutional super-resolution hot path (SRCNN / ESPCN / FSRCNN / VDSR / EDSR /
LapSRN / SRGAN on the shared base_networks blocks), drop-in for the nn.Module surface of
togheppi/pytorch-super-resolution-model-collection.  See DESIGN.md."""
from . import _lib, ops, layers, base_networks, utils, models, optim, dp, trainers, data  # noqa: F401
from .models import (SRCNNNet, ESPCNNet, FSRCNNNet, VDSRNet, EDSRNet, LapSRNNet, SRGANGenerator,  # noqa: F401
                     SRGANDiscriminator, FeatureExtractor)

__all__ = ["ops", "layers", "base_networks", "utils", "models", "SRCNNNet", "ESPCNNet", "FSRCNNNet", "VDSRNet",
           "EDSRNet", "LapSRNNet", "SRGANGenerator", "SRGANDiscriminator", "FeatureExtractor"]
