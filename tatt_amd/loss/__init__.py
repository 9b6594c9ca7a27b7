from .image_loss import ImageLoss  # noqa: F401
