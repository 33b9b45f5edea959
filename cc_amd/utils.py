"""Checkpoint wire format of the reference (utils.py:55-63, train.py:257-295,396-413): five files
``{dispnet,posenet,masknet,flownet,optimizer}_checkpoint.pth.tar`` each holding ``{'epoch', 'state_dict'}``, copied to
``*_model_best.pth.tar`` when ``is_best``.  Checkpoints written here load in the reference with its own
``torch.load`` + ``load_state_dict`` lines and vice versa (the state_dict keys are identical, tests/test_api_contract.py).
The logging helpers of utils.py (``tensor2array``, colormaps) are out of scope."""
import os
import shutil

import torch

FILE_PREFIXES = ['dispnet', 'posenet', 'masknet', 'flownet', 'optimizer']      # utils.py:56


def save_checkpoint(save_path, dispnet_state, posenet_state, masknet_state, flownet_state, optimizer_state, is_best,
                    filename='checkpoint.pth.tar'):
    """utils.py:55-63, same signature (save_path: str or path-like)."""
    states = [dispnet_state, posenet_state, masknet_state, flownet_state, optimizer_state]
    for (prefix, state) in zip(FILE_PREFIXES, states):
        torch.save(state, os.path.join(str(save_path), '{}_{}'.format(prefix, filename)))
    if is_best:
        for prefix in FILE_PREFIXES:
            shutil.copyfile(os.path.join(str(save_path), '{}_{}'.format(prefix, filename)),
                            os.path.join(str(save_path), '{}_model_best.pth.tar'.format(prefix)))


def load_pretrained(net, path, map_location=None):
    """train.py:257-284: ``weights = torch.load(path); net.load_state_dict(weights['state_dict'])``."""
    weights = torch.load(path, map_location=map_location)
    net.load_state_dict(weights['state_dict'])
    return weights.get('epoch')


def resume(save_path, disp_net, pose_net, mask_net, flow_net, map_location=None):
    """train.py:286-295: restore the four networks from ``*_checkpoint.pth.tar``.  -> epoch of the checkpoint.
    Call it BEFORE building the optimizer / CCTrainer (as train.py does), then `resume_optimizer`."""
    epoch = None
    for prefix, net in zip(FILE_PREFIXES[:4], (disp_net, pose_net, mask_net, flow_net)):
        if net is not None:
            epoch = load_pretrained(net, os.path.join(str(save_path), '{}_checkpoint.pth.tar'.format(prefix)), map_location)
    return epoch


def resume_optimizer(save_path, optimizer, map_location=None):
    """train.py:311-314: ``if (save_path/'optimizer_checkpoint.pth.tar').exists(): optimizer.load_state_dict(...)`` -- Adam
    moments and step count continue where the checkpoint left them.  optimizer: ``CCTrainer.opt`` (FlatAdam), a CCTrainer, or
    a ``torch.optim.Adam`` over the same parameter chain.  -> True when a checkpoint was loaded."""
    path = os.path.join(str(save_path), 'optimizer_checkpoint.pth.tar')
    if not os.path.exists(path):
        return False
    opt = getattr(optimizer, "opt", optimizer)
    opt.load_state_dict(torch.load(path, map_location=map_location)['state_dict'])
    return True
