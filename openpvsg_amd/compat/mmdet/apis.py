import torch


def single_gpu_test(model, data_loader, show=False, out_dir=None, show_score_thr=0.3):
    """[3P] mmdet.apis.single_gpu_test (tools/test.py:245): per batch `model(return_loss=False, rescale=True, **data)`,
    then -- as mmdet 2.25 does -- the instance masks of every result are RLE-encoded (`encode_mask_results`) before the
    result is kept: tuples (bbox, masks) and dicts carrying 'ins_results' (the panoptic test path)."""
    from openpvsg_amd.detectors import encode_mask_results
    model.eval()
    results = []
    for data in data_loader:
        with torch.no_grad():
            result = model(return_loss=False, rescale=True, **data)
        if result and isinstance(result[0], tuple):
            result = [(bbox_results, encode_mask_results(mask_results)) for bbox_results, mask_results in result]
        elif result and isinstance(result[0], dict) and 'ins_results' in result[0]:
            for j in range(len(result)):
                bbox_results, mask_results = result[j]['ins_results']
                result[j]['ins_results'] = (bbox_results, encode_mask_results(mask_results))
        results.extend(result)
    return results


def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=False):
    """One rank's share, then an object all-gather (tools/test.py:252)."""
    import torch.distributed as dist
    part = single_gpu_test(model, data_loader)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return part
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, part)
    if dist.get_rank() != 0:
        return None
    out = []
    for items in zip(*gathered):
        out.extend(items)
    return out[:len(data_loader.dataset)]
