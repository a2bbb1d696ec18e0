from .anchor_free_head import AnchorFreeHead


class MaskFormerHead(AnchorFreeHead):
    pass
