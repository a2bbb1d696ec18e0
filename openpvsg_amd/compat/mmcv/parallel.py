class DataContainer:
    """mmcv.parallel.DataContainer: value + collate hints."""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self._data, self.stack, self.padding_value, self.cpu_only, self.pad_dims = data, stack, padding_value, cpu_only, pad_dims

    @property
    def data(self):
        return self._data

    def __repr__(self):
        return 'DataContainer(%r)' % (self._data,)


class MMDataParallel:
    """Single-device wrapper (tools/test.py:244): moves tensors to the model's device, unwraps DataContainers."""

    def __init__(self, module, device_ids=None, **kw):
        self.module = module

    def eval(self):
        self.module.eval()
        return self

    def _to(self, x, dev):
        import torch
        if isinstance(x, DataContainer):
            x = x.data
            return self._to(x[0] if isinstance(x, list) and len(x) == 1 else x, dev)
        if torch.is_tensor(x):
            return x.to(dev)
        if isinstance(x, list):
            return [self._to(v, dev) for v in x]
        if isinstance(x, tuple):
            return tuple(self._to(v, dev) for v in x)
        if isinstance(x, dict):
            return {k: self._to(v, dev) for k, v in x.items()}
        return x

    def __call__(self, *args, **kwargs):
        import torch
        dev = next(self.module.parameters()).device
        return self.module(*self._to(list(args), dev), **self._to(kwargs, dev))


MMDistributedDataParallel = MMDataParallel
