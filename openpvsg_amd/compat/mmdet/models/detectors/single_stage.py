from openpvsg_amd.blocks import BaseModule


class BaseDetector(BaseModule):
    @property
    def with_neck(self):
        return getattr(self, 'neck', None) is not None


class SingleStageDetector(BaseDetector):
    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x
