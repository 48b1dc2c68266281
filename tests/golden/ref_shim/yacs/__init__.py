"""empty stand-in: GINESignNetPyG/core/config.py imports yacs but the path never uses it."""
