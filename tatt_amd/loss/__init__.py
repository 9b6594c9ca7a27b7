from .image_loss import ImageLoss  # noqa: F401
from .semantic_loss import SemanticLoss  # noqa: F401
