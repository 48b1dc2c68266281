import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
