from canonicalvoting_amd.me.modules.resnet_block import BasicBlock  # noqa: F401
