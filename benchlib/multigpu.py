"""What only the N > 1 runs of bench.py do besides the timed steps."""
import json
import os
import time

def multi_gpu_extras(args, torch, dist, femto_amd, ix, batch, rank, world, local_rank, dev, backend, native, payload_of, index_path,
                     elapsed, npats, per_rank):
    """N > 1 only, after the timed run: the step again (a) without any gather and (b) with the OTHER gather implementation, so
    one driver run tells search time from gather time and the two gathers apart.  Nothing here has ever run on more than one
    physical GPU before the driver's scaling run, so every part is guarded: a watchdog prints a minimal result line (the
    headline value and what is known so far) and ends the process if a part does not finish -- the headline never depends
    on an extra."""
    import threading
    out = {}
    state = {"phase": "start"}

    def bail():
        if rank == 0:
            line = {"metric": "patterns/sec (count+locate) on " + ("1 GiB index" if args.text_log2 == 30 else f"2^{args.text_log2} B index"),
                    "value": world * npats * args.steps / elapsed, "unit": "patterns/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                    "config": {"workload": f"{npats} patterns per GPU, count()+locate(max_occs={args.max_occs}), layout {args.layout}",
                               "per_rank": per_rank, "note": f"multi-GPU extra '{state['phase']}' did not finish within its time limit: "
                                                             "minimal line, no roofline / cpu_baseline"},
                    "roofline": None, "cpu_baseline": None, "extra": dict(out, timeout=state["phase"])}
            print(json.dumps(line), flush=True)
        os._exit(0)

    def timed(name, fn, steps=5, limit_s=90):
        state["phase"] = name
        t = threading.Timer(limit_s, bail)
        t.daemon = True
        t.start()
        try:
            fn()                                   # warm-up
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            dt = time.perf_counter() - t0
            tm = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            out[name] = {"value": world * npats * steps / float(tm.item()), "unit": "patterns/s", "ms_per_step": 1e3 * float(tm.item()) / steps,
                         "steps": steps}
        except Exception as ex:      # noqa: BLE001
            out[name] = {"error": repr(ex)}
        finally:
            t.cancel()

    stream = torch.cuda.current_stream().cuda_stream
    timed("search_only_no_gather", lambda: batch.step(ix, args.max_occs, stream, 0))
    if backend == "nccl":
        payload = payload_of(0)
        nbytes = payload.numel() * payload.element_size()
        if native:      # the timed run used femto_amd_comm_gather: now torch.distributed.gather (RCCL)
            lists = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None

            def other():
                batch.step(ix, args.max_occs, stream, 0)
                dist.gather(payload_of(0), lists, dst=0)
            timed("gather_torch_distributed", other)
        else:           # the timed run used torch.distributed.gather: now the C ABI's grouped ncclSend / ncclRecv
            try:
                state["phase"] = "native_comm_init"
                t = threading.Timer(90, bail)
                t.daemon = True
                t.start()
                ids = [femto_amd.Index.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ix.comm_init(ids[0], world, rank)
                t.cancel()
                recv = torch.empty((world,) + tuple(payload.shape), dtype=payload.dtype, device=dev) if rank == 0 else None

                def other():
                    batch.step(ix, args.max_occs, stream, 0)
                    p = payload_of(0)
                    ix.comm_gather(p.data_ptr(), recv.data_ptr() if rank == 0 else 0, nbytes, 0, stream)
                timed("gather_native_ncclSendRecv", other)
                out["native_comm"] = ix.comm_info()
            except Exception as ex:      # noqa: BLE001
                out["gather_native_ncclSendRecv"] = {"error": repr(ex)}
        out["gather_payload_bytes_per_rank"] = int(nbytes)
    state["phase"] = "done"
    return out if rank == 0 else None
