"""Drop-in for the reference's ``models`` package (models/__init__.py): ``getattr(models, name)(...)`` as at
train.py:245-255.  Hot-path nets (BASELINE.json): DispResNet6, PoseNetB6, MaskNet6, Back2Future; plus the
config-1 baselines DispNetS and PoseExpNet, and the alternative architectures train.py:84-91 can select (SURVEY.md 8f rank 4:
DispNetS6, DispResNetS6, PoseNet6, MaskResNet6, FlowNetC6)."""
from .back2future import Model as Back2Future
from .DispNetS import DispNetS
from .DispResNet6 import DispResNet6
from .MaskNet6 import MaskNet6
from .PoseExpNet import PoseExpNet
from .PoseNetB6 import PoseNetB6
from .DispNetS6 import DispNetS6
from .DispResNetS6 import DispResNetS6
from .MaskResNet6 import MaskResNet6
from .PoseNet6 import PoseNet6
from .FlowNetC6 import FlowNetC6
