from . import pytorch
from .pytorch.glob import SetTransformerEncoder
