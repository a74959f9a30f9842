"""Bisection of the capture_end abort (see capture_diag.py): each MODE is one process.
  fwd        two-stream IRN passes with grad ENABLED inside a capture, no backward
  bwd        + backward (the aborting configuration)
  bwd1t      + backward with the autograd engine single-threaded (backward runs on the capturing thread)
  bwdkeep    + backward, python-level events of wait_stream kept alive until the capture has ended
"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SININN_IRN_HG_TRAIN'] = '1'
import torch


def main():
    mode = sys.argv[1]
    import lit_wrapper
    from sin_inn_amd import irn, modules
    from sin_inn_amd.modules import join_capturing_helpers, join_side_streams, side_stream_if_any
    torch.manual_seed(5)
    opt = types.SimpleNamespace(scale=4, num_coupling=2, lr_window=2, architecture='IRN', gpu_ids=[0], rotation=5.0, translation=5.0,
                                tcr_iters=1, lambda_fwd_rec=1.0, lambda_fwd_mmd=0.0, lambda_latent_nll=0.0, lambda_bwd_rec=1.0,
                                lambda_bwd_mmd=0.0, lambda_bwd_tcr=0.0, learning_rate=1e-4, adam_betas=[0.9, 0.99], weight_decay=1e-5,
                                temp=0.8, operation='train', fps=1, lr_dims=20, z_dims=172, precision='fp32', hip_graph=False)
    model = lit_wrapper.SingleVideoINN(3, 64, 64, opt).cuda()
    for m in model.inn.modules():
        if isinstance(m, irn.DenseBlock):
            torch.nn.init.normal_(m.conv5.weight, std=0.02)
    model.attach_optimizer()
    if mode == 'bwd1t':
        torch.autograd.set_multithreading_enabled(False)
    keep = []
    if mode == 'bwdkeep':
        real = torch.cuda.Stream.wait_stream

        def wait_stream(self, other):
            ev = other.record_event()
            keep.append(ev)
            self.wait_event(ev)
        torch.cuda.Stream.wait_stream = wait_stream
    hr = torch.rand(4, 64, 64, 3, device='cuda').permute(0, 3, 1, 2)

    def work():
        out = model.inn(hr)
        if mode != 'fwd':
            out.square().mean().backward()
        return out
    for _ in range(2):
        work()                       # eager warm-up: packs, allocator
    join_side_streams()
    torch.cuda.synchronize()
    print(f'[diag2 {mode}] capturing', flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='relaxed'):
        out = work()
        side = side_stream_if_any(hr.device)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        print(f'[diag2 {mode}] unjoined at the end: {join_capturing_helpers()}', flush=True)
    print(f'[diag2 {mode}] capture ended', flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(f'[diag2 {mode}] replay ok, |out| = {float(out.norm()):.4f}', flush=True)


if __name__ == '__main__':
    main()
