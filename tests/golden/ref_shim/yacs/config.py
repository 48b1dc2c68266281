class CfgNode(dict):
    pass
