from openpvsg_amd.blocks import BaseModule


class BaseDenseHead(BaseModule):
    pass


class AnchorFreeHead(BaseDenseHead):
    """Base class name the reference heads inherit from; they call `super(AnchorFreeHead, self).__init__(init_cfg)`."""
