"""Stand-in package for torch_geometric==2.0.1 (absent): only the ops the reference path calls."""
