"""Diagnostic for the capture_end abort of round 3 (gpurun_out/r03w_tests.log): captures one IRN training step with the
two-stream InvBlockExp block in DIFFERENTIATED passes (the configuration of commit 6a8be52) and reports, from the graph under
construction, which helper streams hold unjoined work right before the capture ends.  Run as its own process:
    SININN_IRN_HG_TRAIN=1 python tools/capture_diag.py [--no-join]
--no-join reports only (the capture then ends with whatever is unjoined: on ROCm 7.2 that is the abort)."""
import argparse
import ctypes as C
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-join', action='store_true')
    ap.add_argument('--arch', default='IRN')
    ap.add_argument('--no-overlap', action='store_true', help='both pass chains on one stream')
    ap.add_argument('--no-side', action='store_true', help='weight gradients on the pass streams')
    ap.add_argument('--one-thread', action='store_true', help='autograd engine single-threaded')
    ap.add_argument('--keep-events', action='store_true', help='python-level wait_stream events kept alive')
    ap.add_argument('-c', type=int, default=2)
    ap.add_argument('--prefork', default='', help='comma list of helper streams forked straight from the capturing stream before anything else: second,side,aux,aux2')
    ap.add_argument('--no-randn', action='store_true', help='latent from a static buffer (no RNG inside the capture)')
    a = ap.parse_args()
    import lit_wrapper
    from data import FrameStore
    from sin_inn_amd import _lib, modules
    from sin_inn_amd.functional import sample_windows
    torch.manual_seed(5)
    opt = types.SimpleNamespace(scale=4, num_coupling=a.c, lr_window=2, architecture=a.arch, gpu_ids=[0], rotation=5.0, translation=5.0,
                                tcr_iters=1, lambda_fwd_rec=1.0, lambda_fwd_mmd=0.0, lambda_latent_nll=0.0, lambda_bwd_rec=1.0,
                                lambda_bwd_mmd=0.0, lambda_bwd_tcr=0.0, learning_rate=1e-4, adam_betas=[0.9, 0.99], weight_decay=1e-5,
                                temp=0.8, operation='train', fps=1, lr_dims=20, z_dims=172, precision='fp32', hip_graph=True)
    model = lit_wrapper.SingleVideoINN(3, 64, 64, opt).cuda()
    model.attach_optimizer()
    if a.no_overlap:
        model.overlap_passes = False
    if a.no_side:
        modules.USE_SIDE_STREAM[0] = False
    if a.one_thread:
        torch.autograd.set_multithreading_enabled(False)
    keep = []
    if a.keep_events:
        def wait_stream(self, other):
            ev = other.record_event()
            keep.append(ev)
            self.wait_event(ev)
        torch.cuda.Stream.wait_stream = wait_stream
    if a.no_randn:
        zbuf = torch.randn(4, 8, 8, opt.z_dims, device='cuda').permute(0, 3, 1, 2)
        lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: zbuf
    store = FrameStore.synthetic(12, 64, 64).to('cuda')
    g = torch.Generator().manual_seed(7)

    real_join = modules.join_capturing_helpers

    def report():
        cur = torch.cuda.current_stream()
        mine = [(n, s) for n, s in modules.HELPER_STREAMS if s.cuda_stream != cur.cuda_stream]
        handles = (C.c_void_p * len(mine))(*[s.cuda_stream for _, s in mine])
        flags = (C.c_int * len(mine))()
        _lib.check(_lib.lib().sininn_capture_unjoined(C.c_void_p(cur.cuda_stream), handles, len(mine), flags))
        for (n, s), f in zip(mine, flags):
            print(f'[capture_diag] {("not in this capture", "capturing, joined", "capturing, UNJOINED")[f]:22s} {n} ({s.cuda_stream:#x})', flush=True)
        # hip::Stream internals of torch's bundled libamdhip64 (ROCm 7.0 build; offsets read from the disassembly of
        # hip::Stream::EndCapture): +0x2e0 / +0x2e8 begin / end of parallelCaptureStreams_, +0x2a8 parentStream_, +0x2a4 originStream_
        import struct
        names = {s.cuda_stream: n for n, s in modules.HELPER_STREAMS}
        names[cur.cuda_stream] = 'ORIGIN (capturing stream)'
        for h, n in list(names.items()):
            if not h:
                continue
            beg, end = struct.unpack('QQ', C.string_at(h + 0x2e0, 16))
            parent, = struct.unpack('Q', C.string_at(h + 0x2a8, 8))
            origin = C.string_at(h + 0x2a4, 1)[0]
            kids = struct.unpack(f'{(end - beg) // 8}Q', C.string_at(beg, end - beg)) if end > beg else ()
            print(f'[capture_diag] {h:#x} {n}: origin={origin} parent={names.get(parent, hex(parent))} parallelCaptureStreams={[names.get(k, hex(k)) for k in kids]}', flush=True)
        return [] if a.no_join else real_join()
    import sin_inn_amd.modules as m
    m.join_capturing_helpers = report
    if a.prefork:
        from sin_inn_amd import irn
        real_passes = model._passes

        def passes(hr, lr, batch, optim, join=True):
            if torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream()
                second = lit_wrapper._second_stream(hr.device)
                table = {'second': lambda: second, 'side': lambda: modules._side_stream(hr.device), 'aux': lambda: irn._aux_stream(hr.device)}

                def aux2():
                    with torch.cuda.stream(second):
                        return irn._aux_stream(hr.device)
                table['aux2'] = aux2
                for name in a.prefork.split(','):
                    table[name]().wait_stream(cur)
                    print(f'[capture_diag] preforked {name}', flush=True)
            return real_passes(hr, lr, batch, optim, join)
        model._passes = passes
    for i in range(6):
        idx = torch.randint(2, 10, (4,), generator=g).cuda()
        hr, lr = sample_windows(store.hr, store.lr, idx, 2)
        print(f'[capture_diag] step {i}', flush=True)
        model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)
    torch.cuda.synchronize()
    captured = any('graph' in v for v in model.__dict__.get('_graphs', {}).values())
    print(f'[capture_diag] done: captured={captured} loss={float(model._logged["train"]):.6f} loose={model.__dict__.get("_capture_loose")}', flush=True)


if __name__ == '__main__':
    main()
