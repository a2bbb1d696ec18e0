from openpvsg_amd.blocks import BaseModule


class BasePanopticFusionHead(BaseModule):
    def __init__(self, num_things_classes=80, num_stuff_classes=53, test_cfg=None, loss_panoptic=None,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.test_cfg = test_cfg
