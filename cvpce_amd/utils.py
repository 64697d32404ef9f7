"""The hot-path members of /root/reference/cvpce/utils.py (plotting helpers are out of scope)."""
import re


def scale_to_tanh(tensor):
    """utils.py:280-281"""
    return tensor * 2 - 1


def scale_from_tanh(tensor):
    """utils.py:283-284"""
    return (tensor + 1) / 2


def trim_module_prefix(state_dict):
    """utils.py:276-278: strip the DDP `module.` prefix from checkpoint keys."""
    regex = re.compile(r'^module\.(.*)$')
    return {regex.match(k).group(1): v for k, v in state_dict.items()}
