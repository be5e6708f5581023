from torch import nn

Model = nn.Module  # the reference's Model wrapper is not on the semi-supervised path
