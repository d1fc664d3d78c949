"""Small host helpers of the reference that sit on the hot path's boundary
(utils/utils.py:148-197): checkpoint loading by name and the 5-D -> 4-D crop flatten."""
import torch


def load_weights(model, pretrained_dict, multi_gpu=False, by_name=False):
    """utils/utils.py:148-192: update ``model``'s state_dict from ``pretrained_dict``.
    ``by_name``: keep only keys that exist in the model with the same shape (others are dropped
    with a warning).  ``multi_gpu``: the model is wrapped (``.module``)."""
    target = model.module if multi_gpu else model
    model_dict = target.state_dict()
    pretrained_dict = dict(pretrained_dict)
    for k in [k for k in pretrained_dict if 'tsm_shift' in k]:
        pretrained_dict[k.replace('tsm_shift', 'tsm_buffer')] = pretrained_dict[k]
    if by_name:
        for k in list(pretrained_dict):
            if k not in model_dict:
                print(f'Warning:  "{k}" is not exist and has been deleted!!')
                del pretrained_dict[k]
            elif model_dict[k].shape != pretrained_dict[k].shape:
                print(f'Warning:  "{k}":{pretrained_dict[k].shape}->{model_dict[k].shape}')
                del pretrained_dict[k]
    model_dict.update(pretrained_dict)
    target.load_state_dict(model_dict)
    return model


def pkl_convert(param):
    """utils/utils.py:141-146: strip the DataParallel ``module.`` prefix."""
    return {k.replace('module.', ''): v for k, v in param.items() if 'module.' in k}


def tensor_dim5to4(tensor):
    """utils/utils.py:194-197: [batch, crops, C, H, W] -> [batch*crops, C, H, W]."""
    b, crops, c, h, w = tensor.shape
    return tensor.reshape(b * crops, c, h, w)
