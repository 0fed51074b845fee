#!/usr/bin/env python3
"""S2E_DETERMINISTIC=1: two trainers built from the same weights and fed the same batches end with bit-identical parameter arenas
(prints one JSON line; run in a fresh process -- the library reads the switch once).
    S2E_DETERMINISTIC=1 python tools/check_deterministic.py [--ngf 32] [--iters 3] [--graphs]"""
import argparse, contextlib, io, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ngf', type=int, default=32)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--graphs', action='store_true')
    args = ap.parse_args()
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
    dev = torch.device('cuda', 0)
    batches = [bench.make_data(args.batch, 256, 1234 + i, dev) for i in range(2)]
    arenas = []
    for run in range(2):
        opt = default_opt(ngf=args.ngf, ndf=args.ngf, crop_size=256, aspect_ratio=1.0, batchSize=args.batch, compute_dtype='bf16',
                          gpu_ids=[0], hip_graphs=args.graphs)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Pix2PixTrainer(opt)
        bench.fill_weights(tr.pix2pix_model)
        for it in range(args.iters):
            b = batches[it % 2]
            tr.run_generator_one_step(dict(b)); tr.run_discriminator_one_step(dict(b))
        torch.cuda.synchronize()
        arenas.append((tr.optimizer_G.flat_p.detach().clone(), tr.optimizer_D.flat_p.detach().clone()))
        names = []
        for tag, net in (('G', tr.pix2pix_model.netG), ('E', tr.pix2pix_model.netE), ('D', tr.pix2pix_model.netD)):
            opt_ = tr.optimizer_D if tag == 'D' else tr.optimizer_G
            off = {id(p): o for p, o in zip(opt_.params, opt_.offsets)}
            for n_, p_ in net.named_parameters():
                if id(p_) in off:
                    names.append((tag == 'D', off[id(p_)], p_.numel(), tag + '.' + n_))
        del tr
    dg = float((arenas[0][0] - arenas[1][0]).abs().max()); dd = float((arenas[0][1] - arenas[1][1]).abs().max())
    differing = [nm for isd, o, n_, nm in names if not torch.equal(arenas[0][1 if isd else 0][o:o + n_], arenas[1][1 if isd else 0][o:o + n_])]
    print(json.dumps({'differing_parameters': differing[:12], 'n_differing': len(differing), 'deterministic_env': os.environ.get('S2E_DETERMINISTIC', '0'), 'iters': args.iters, 'graphs': args.graphs,
                      'max_abs_diff_G': dg, 'max_abs_diff_D': dd,
                      'bit_identical': bool(torch.equal(arenas[0][0], arenas[1][0]) and torch.equal(arenas[0][1], arenas[1][1]))}))


if __name__ == '__main__':
    main()
