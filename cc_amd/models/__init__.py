"""Drop-in for the reference's ``models`` package (models/__init__.py): ``getattr(models, name)(...)`` as at
train.py:245-255.  Hot-path nets (BASELINE.json): DispResNet6, PoseNetB6, MaskNet6, Back2Future; plus the
config-1 baselines DispNetS and PoseExpNet."""
from .back2future import Model as Back2Future
from .DispNetS import DispNetS
from .DispResNet6 import DispResNet6
from .MaskNet6 import MaskNet6
from .PoseExpNet import PoseExpNet
from .PoseNetB6 import PoseNetB6
