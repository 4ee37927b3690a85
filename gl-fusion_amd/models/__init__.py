"""Host-side mirror of the reference's ``models`` package for the hot path: same module
names, constructor signatures, forward contracts and state_dict keys
(GLfusion/models/{ours,segmentation,deeplabv3,_utils}.py), every forward running on the HIP
engine."""
from .ours import Foreground_and_Background, Global_and_Local, Global_and_Local_Temporal, Global_and_Local_conv_merge, Global_and_Local_cyc_nofusion, Global_only, Global_only_cyc_nofusion, Local_only, TPAVIModule, model19, Global_and_Local_CPS  # noqa: F401
from .segmentation import deeplabv3_resnet50_iekd  # noqa: F401
from .deeplabv3 import ASPP, DeepLabHead  # noqa: F401
