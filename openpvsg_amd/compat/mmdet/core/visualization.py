from openpvsg_amd.compat._policy import _training_only

imshow_det_bboxes = _training_only('imshow_det_bboxes')
