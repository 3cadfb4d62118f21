"""Weights-directory naming contract + INIT_LEN helper of the reference
(code/utilities/utilities.py:872-919; the directory name is written by code/train.py:103)."""
import re


def parse_hidden_size(model_name):
    """'...-HS[64]-...' -> 64   (code/utilities/utilities.py:872-884)."""
    return int(re.search(r'-HS\[(.\d*)\]-', model_name).group(1))


def parse_model(model_name):
    """Text before the first '-'   (code/utilities/utilities.py:887-899)."""
    return model_name[:model_name.find('-')]


def parse_loss(model_name):
    """'...]-L[DCPreESR]-DS[...' -> 'DCPreESR'   (code/utilities/utilities.py:902-914)."""
    return re.search(r'\]-L\[(.*)\]-DS\[', model_name).group(1)


def nextpow2(number):
    """Next power of two >= number   (code/utilities/utilities.py:917-919)."""
    return 2**(number - 1).bit_length()
