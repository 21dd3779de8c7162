"""Overlay module for the reference's `layers.py`: same class names and parameter names, CSR
aggregation on HIP.  `from layers import *` also hands the caller what the reference module
re-exported (layers.py:1-9)."""
import math  # noqa: F401
import random  # noqa: F401

import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from torch import nn as nn  # noqa: F401
from torch.nn.modules.module import Module  # noqa: F401
from torch.nn.parameter import Parameter  # noqa: F401

from geometrics_amd.layers import (BatchGCNMax, BatchZERON_GCN, Batch_Image_ZERON_GCNGCN,  # noqa: F401
                                   GCNMax, ZERON_GCN)
