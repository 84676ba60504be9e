"""a data-parallel FlowNet2C training / inference harness around the three HIP layers -- one process per
GPU over torch.distributed (RCCL on MI355X), replacing the reference's single-process nn.DataParallel loop (main.py:187-201,
:246-340).  The convolution stack is plain torch.nn (MIOpen); what this repo adds is the Correlation layer, the fused
multi-scale loss and the gradient exchange."""
