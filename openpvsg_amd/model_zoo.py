"""Model-section configs of the two R50 networks the reference ships, generated from their
hyper-parameters (tests/test_config_compat.py checks, in the build container, that they are equal
to what `Config.fromfile` yields for the reference's own config files:
  configs/mask2former/mask2former_r50_lsj_8x2_50e_coco-panoptic_custom_single_video_test.py
  configs/mask2former_vps/mask2former_video_r50_single_video_test.py).
Used by bench.py / smoke() / the GPU tests, which cannot read /root/reference."""

NUM_THINGS, NUM_STUFF = 115, 11


def _encoder_layer(dims, heads, levels, points, ffn):
    return dict(
        type='BaseTransformerLayer',
        attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=dims, num_heads=heads,
                       num_levels=levels, num_points=points, im2col_step=64, dropout=0.0,
                       batch_first=False, norm_cfg=None, init_cfg=None),
        ffn_cfgs=dict(type='FFN', embed_dims=dims, feedforward_channels=ffn, num_fcs=2, ffn_drop=0.0,
                      act_cfg=dict(type='ReLU', inplace=True)),
        operation_order=('self_attn', 'norm', 'ffn', 'norm'))


def _decoder_layer(dims, heads, ffn):
    return dict(
        type='DetrTransformerDecoderLayer',
        attn_cfgs=dict(type='MultiheadAttention', embed_dims=dims, num_heads=heads, attn_drop=0.0,
                       proj_drop=0.0, dropout_layer=None, batch_first=False),
        ffn_cfgs=dict(embed_dims=dims, feedforward_channels=ffn, num_fcs=2,
                      act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0.0, dropout_layer=None,
                      add_identity=True),
        feedforward_channels=ffn,
        operation_order=('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm'))


def _losses(num_classes):
    return dict(
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0, reduction='mean',
                      class_weight=[1.0] * num_classes + [0.1]),
        loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='mean', loss_weight=5.0),
        loss_dice=dict(type='DiceLoss', use_sigmoid=True, activate=True, reduction='mean',
                       naive_dice=True, eps=1.0, loss_weight=5.0))


def panoptic_head_cfg(video=False, dims=256, queries=100, heads=8, levels=3):
    pe = 'SinePositionalEncoding3D' if video else 'SinePositionalEncoding'
    cfg = dict(
        type='Mask2FormerVideoHead' if video else 'Mask2FormerHeadCustom',
        in_channels=[256, 512, 1024, 2048], strides=[4, 8, 16, 32], feat_channels=dims,
        out_channels=dims, num_things_classes=NUM_THINGS, num_stuff_classes=NUM_STUFF,
        num_queries=queries, num_transformer_feat_level=levels,
        pixel_decoder=dict(
            type='MSDeformAttnPixelDecoder', num_outs=3, norm_cfg=dict(type='GN', num_groups=32),
            act_cfg=dict(type='ReLU'),
            encoder=dict(type='DetrTransformerEncoder', num_layers=6,
                         transformerlayers=_encoder_layer(dims, heads, levels, 4, 4 * dims), init_cfg=None),
            positional_encoding=dict(type='SinePositionalEncoding', num_feats=dims // 2, normalize=True),
            init_cfg=None),
        enforce_decoder_input_project=False,
        positional_encoding=dict(type=pe, num_feats=dims // 2, normalize=True),
        transformer_decoder=dict(type='DetrTransformerDecoder', return_intermediate=True, num_layers=9,
                                 transformerlayers=_decoder_layer(dims, heads, 8 * dims), init_cfg=None),
        **_losses(NUM_THINGS + NUM_STUFF))
    if video:
        cfg['loss_sem_seg'] = None
    return cfg


def _train_cfg():
    return dict(
        num_points=12544, oversample_ratio=3.0, importance_sample_ratio=0.75,
        assigner=dict(type='MaskHungarianAssigner', cls_cost=dict(type='ClassificationCost', weight=2.0),
                      mask_cost=dict(type='CrossEntropyLossCost', weight=5.0, use_sigmoid=True),
                      dice_cost=dict(type='DiceCost', weight=5.0, pred_act=True, eps=1.0)),
        sampler=dict(type='MaskPseudoSampler'))


def mask2former_r50_model_cfg(video=False):
    test_cfg = dict(panoptic_on=True, semantic_on=False, instance_on=True, max_per_image=100, iou_thr=0.8,
                    filter_low_score=True)
    if video:
        test_cfg.update(object_mask_thr=0.8)
    test_cfg.update(return_query=True)
    fusion = dict(type='MaskFormerFusionHeadCustom', num_things_classes=NUM_THINGS,
                  num_stuff_classes=NUM_STUFF, loss_panoptic=None, init_cfg=None)
    return dict(
        type='Mask2FormerVideoCustom' if video else 'Mask2FormerCustom',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                      norm_cfg=dict(type='SyncBN', requires_grad=True) if video
                      else dict(type='BN', requires_grad=False),
                      norm_eval=True, style='pytorch',
                      init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50')),
        panoptic_head=panoptic_head_cfg(video),
        panoptic_fusion_head=fusion, train_cfg=_train_cfg(), test_cfg=test_cfg, init_cfg=None)
