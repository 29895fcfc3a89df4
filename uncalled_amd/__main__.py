"""`python -m uncalled_amd {index,map,sim,pafstats}` -- the subcommands of the reference's `scripts/uncalled` that do not need
a sequencer (index_cmd :38-78, map_cmd :126-167, realtime_cmd :170-256 fed from fast5 files, pafstats) with the same options (uncalled/args.py:90-124,244-304) on the GPU path.
"""
import argparse
import os
import sys
import time

MAX_SLEEP = 0.01
BWA_SUFFS = (".amb", ".ann", ".bwt", ".pac", ".sa")


def _assert_exists(fname):
    if not os.path.exists(fname):
        sys.stderr.write("Error: '%s' does not exist\n" % fname)
        sys.exit(1)


def load_fast5s(paths, recursive):
    """scripts/uncalled:80-119: directories (optionally recursive), .fast5 files, or text files of file names."""
    def keep(path):
        if path.startswith("#") or not path.endswith("fast5"):
            return None
        path = os.path.abspath(path)
        if not os.path.isfile(path):
            sys.stderr.write("Warning: \"%s\" is not a fast5 file.\n" % path)
            return None
        return path

    for path in paths:
        path = path.strip()
        if not os.path.exists(path):
            sys.stderr.write("Error: \"%s\" does not exist\n" % path)
            sys.exit(1)
        if os.path.isdir(path):
            if recursive:
                for root, _, files in os.walk(path):
                    for f in files:
                        yield keep(os.path.join(root, f))
            else:
                for f in os.listdir(path):
                    yield keep(os.path.join(path, f))
        elif path.endswith(".fast5"):
            yield keep(path)
        else:
            with open(path) as fh:
                for line in fh:
                    yield keep(line.strip())


def index_cmd(args):
    from . import capi, index_params
    prefix = args.bwa_prefix or args.fasta_filename
    if all(os.path.exists(prefix + s) for s in BWA_SUFFS):
        sys.stderr.write("Using previously built BWA index.\nNote: to fully re-build the index delete files with the "
                         "\"%s.*\" prefix.\n" % prefix)
    else:
        # the suffix sort inside bwa_idx_build (bwa_index.hpp:92-101) runs on the GPU behind the C ABI (unc_build_suffix_array: no torch
        # in the process) for references of fewer than 2^30 bases; larger ones go through the chunked builder (torch tensors around the
        # same radix sort)
        from . import build_index
        names, annos, seqs = build_index.read_fasta(args.fasta_filename)
        if 2 * sum(len(s) for s in seqs) < (1 << 31):
            codes, holes, n_ambs = build_index.encode_contigs(seqs)
            build_index.build_from_codes(prefix, names, annos, [len(s) for s in seqs], codes, holes, n_ambs, verbose=True,
                                         sa_device="cuda:%d" % args.device)
        else:
            from . import build_index_big
            codes, holes, n_ambs = build_index.encode_contigs(seqs)
            build_index_big.build_from_codes_big(prefix, names, annos, [len(s) for s in seqs], codes, holes, n_ambs, device="cuda:%d" % args.device)
    sys.stderr.write("Initializing parameter search\n")
    presets = [("default", dict(tgt_speed=115))]
    for tgt in (args.probs.split(",") if args.probs else []):
        presets.append(("prob_%s" % tgt, dict(tgt_prob=float(tgt))))
    for tgt in (args.speeds.split(",") if args.speeds else []):
        presets.append(("speed_%s" % tgt, dict(tgt_speed=float(tgt))))
    # the loader wants a preset (self-alignment does not read the thresholds): a placeholder .uncl lives only for the
    # duration of the search and never survives a failed or interrupted one -- `map` would load it silently
    placeholder = not os.path.exists(prefix + ".uncl")
    if placeholder:
        with open(prefix + ".uncl", "w") as f:
            f.write("default\t-10.0\t0.00000\t0.000\n")
    done = False
    try:
        ix = capi.Index(prefix, device=args.device)
        index_params.parameterize(ix, prefix, presets=presets, max_sample_dist=args.max_sample_dist,
                                  min_samples=args.min_samples, max_samples=args.max_samples, kmer_len=args.kmer_len,
                                  matchpr1=args.matchpr1, matchpr2=args.matchpr2,
                                  pathlen_percentile=args.pathlen_percentile, max_replen=args.max_replen)
        done = True
    finally:
        if placeholder and not done and os.path.exists(prefix + ".uncl"):
            os.unlink(prefix + ".uncl")
    sys.stderr.write("Done\n")


def shard_files(files, n_shards):
    """fast5 files of shard i = every n-th file starting at i: files (reads) are independent units, no exchange"""
    files = [f for f in files if f is not None]
    return [files[i::n_shards] for i in range(n_shards)]


def worker_cmd(args, list_name, dev):
    """command line of one `--gpus N` worker: everything but `-n`, which is a cap on the whole job (see map_multi_gpu)"""
    cmd = [sys.executable, "-m", "uncalled_amd", "map", args.bwa_prefix, list_name, "--device", str(dev), "--gpus", "1"]
    for opt, val in (("-p", args.idx_preset), ("-l", args.read_list), ("-e", args.max_events),
                     ("-c", args.max_chunks), ("--chunk-time", args.chunk_time), ("--batch-reads", args.batch_reads)):
        if val is not None:
            cmd += [opt, str(val)]
    return cmd


def map_multi_gpu(args, argv, make_cmd=worker_cmd):
    """`map --gpus N`: one worker process per GPU (index replicated, fast5 files dealt round-robin), PAF lines of all
    workers forwarded to stdout as they come.  SURVEY 8(e): no collective, no data-path communication.
    `-l` goes to every worker in full (a worker only sees its own files, so each read id matches in one of them);
    `-n` is a cap on the whole job and is applied HERE, on the forwarded lines: workers are stopped once it is reached."""
    import subprocess
    import tempfile
    import threading
    shards = [x for x in shard_files(list(load_fast5s(args.fast5s, args.recursive)), args.gpus) if x]
    procs, lists = [], []
    lock = threading.Lock()
    state = {"n": 0, "full": False}
    limit = args.max_reads

    def pump(p):
        for line in p.stdout:
            with lock:
                if limit is not None and state["n"] >= limit:
                    state["full"] = True
                    break
                sys.stdout.write(line)
                state["n"] += 1
        p.stdout.close()

    try:
        for dev, files in enumerate(shards):
            lst = tempfile.NamedTemporaryFile("w", suffix=".fast5s.txt", delete=False)
            lists.append(lst.name)
            lst.write("\n".join(files) + "\n")
            lst.close()
            # UNC_PIN_WORLD: the worker keeps its host threads (HDF5 loader, staging, PAF writer) on its GPU's NUMA node
            procs.append(subprocess.Popen(make_cmd(args, lst.name, dev), stdout=subprocess.PIPE, text=True, bufsize=1,
                                          env={**os.environ, "UNC_PIN_WORLD": str(len(shards))}))
        threads = [threading.Thread(target=pump, args=(p,)) for p in procs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if state["full"]:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        codes = [p.wait() for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for name in lists:
            if os.path.exists(name):
                os.unlink(name)
    sys.stdout.flush()
    if any(codes) and not state["full"]:
        sys.stderr.write("Error: worker exit codes %s\n" % codes)
        sys.exit(1)


def map_cmd(args):
    if getattr(args, "gpus", 1) > 1:
        return map_multi_gpu(args, sys.argv)
    from . import _uncalled_amd as unc
    conf = unc.Conf()
    for k, v in vars(args).items():
        if v is not None and not k.startswith("_") and hasattr(conf, k):
            setattr(conf, k, v)
    _assert_exists(conf.bwa_prefix + ".bwt")
    _assert_exists(conf.bwa_prefix + ".uncl")
    if conf.read_list:
        _assert_exists(conf.read_list)
    pin_world = int(os.environ.get("UNC_PIN_WORLD", "0") or 0)
    if pin_world > 1:
        from .numa import pin_to_gpu_node
        sys.stderr.write("GPU %d: host threads -> %s\n" % (args.device, pin_to_gpu_node(args.device, pin_world)))
    mapper = unc.MapPool(conf)
    sys.stderr.write("Loading fast5s\n")
    for f in load_fast5s(args.fast5s, args.recursive):
        if f is not None:
            mapper.add_fast5(f)
    sys.stderr.write("Mapping\n")
    sys.stderr.flush()
    try:
        while mapper.running():
            t0 = time.time()
            for p in mapper.update():
                p.print_paf()
            dt = time.time() - t0
            if dt < MAX_SLEEP:
                time.sleep(MAX_SLEEP - dt)
    except KeyboardInterrupt:
        pass
    sys.stderr.write("Finishing\n")
    mapper.stop()


class _Channel:
    """What the decision session remembers about one channel."""
    __slots__ = ("last_chunk_at", "ejected_read")

    def __init__(self, now):
        self.last_chunk_at = now        # when the channel's newest chunk was handed to the pool (decision latency is taken from it)
        self.ejected_read = None        # number of the read this channel was told to eject: its late chunks are dropped


class RealtimeSession:
    """ReadUntil decisions over a RealtimePool and a chunk source with the ClientSim surface: the behaviour of the reference's
    `uncalled realtime / sim` main loop (scripts/uncalled:216-256 -- the caller of this repo's boundary, not part of it),
    written as a table of verdicts instead of a branch ladder.

    verdict of a result  | PAF tag (seconds since the channel's last chunk) | told to the chunk source
    ---------------------+--------------------------------------------------+--------------------------------
    "ended"              | Paf.ENDED                                        | stop_receiving_read
    "eject"              | Paf.EJECT (+ Paf.DELAY = the source's answer)    | unblock_read, later chunks of that read dropped
    "keep"               | Paf.KEEP                                         | stop_receiving_read
    A mapped read is ejected when depleting, an unmapped one when enriching; everything else is kept."""

    def __init__(self, unc, conf, client, pool, emit=None, clock=time.time, sim=True):
        self.unc, self.client, self.pool, self.clock = unc, client, pool, clock
        self.sim = sim      # the chunk source is the simulator: an ejection's answer is a delay, written as Paf.DELAY (scripts/uncalled:231-232)
        self.emit = emit or (lambda paf: paf.print_paf())
        self.eject_mapped = conf.realtime_mode == int(unc.RealtimePool.DEPLETE)
        self.skip_odd = conf.active_chs == int(unc.RealtimePool.EVEN)
        self.deadline = conf.duration * 3600 if conf.duration else None
        now = clock()
        self.chan = {c: _Channel(now) for c in range(1, conf.num_channels + 1)}
        self.act = {"ended": self._ended, "eject": self._eject, "keep": self._keep}

    def verdict(self, paf):
        if paf.is_ended():
            return "ended"
        return "eject" if bool(paf.is_mapped()) == self.eject_mapped else "keep"

    def _ended(self, ch, number, paf, waited):
        paf.set_float(self.unc.Paf.ENDED, waited)
        self.client.stop_receiving_read(ch, number)

    def _keep(self, ch, number, paf, waited):
        paf.set_float(self.unc.Paf.KEEP, waited)
        self.client.stop_receiving_read(ch, number)

    def _eject(self, ch, number, paf, waited):
        paf.set_float(self.unc.Paf.EJECT, waited)
        answer = self.client.unblock_read(ch, number)
        if self.sim:
            paf.set_int(self.unc.Paf.DELAY, answer)
        self.chan[ch].ejected_read = number

    def decide(self):
        """every result the pool has ready -> verdict -> PAF line"""
        for ch, number, paf in self.pool.update():
            self.act[self.verdict(paf)](ch, number, paf, self.clock() - self.chan[ch].last_chunk_at)
            self.emit(paf)

    def feed(self):
        """the source's new chunks -> the pool (channels that are switched off and ejected reads aside)"""
        for ch, chunk in self.client.get_read_chunks():
            if self.skip_odd and ch % 2:
                self.client.stop_receiving_read(ch, chunk.number)
            elif self.chan[ch].ejected_read == chunk.number:
                # (the reference's own words, spelling included: the line is part of the PAF stream its consumers see, scripts/uncalled:250)
                sys.stdout.write("# recieved chunk from %s after unblocking\n" % chunk.id)
            else:
                self.chan[ch].last_chunk_at = self.clock()
                self.pool.add_chunk(chunk)

    def run(self, sleep=time.sleep):
        """decide / feed in ticks of MAX_SLEEP until the source has stopped and nothing is left in the pool, or the
        run's duration is over"""
        while self.client.is_running or not self.pool.all_finished():
            tick = self.clock()
            self.decide()
            if not self.client.is_running:
                return          # the source ran dry: what is still undecided stays so
            self.feed()
            if self.deadline is not None and self.client.get_runtime() >= self.deadline:
                return
            spare = MAX_SLEEP - (self.clock() - tick)
            if spare > 0:
                sleep(spare)


def realtime_loop(unc, conf, client, pool, sim=True, emit=None, sleep=time.sleep):
    """`uncalled sim`'s main loop: a RealtimeSession run to its end.  `sim`: the chunk source is the simulator (the MinKNOW client,
    the only other one, is out of scope): its answer to an ejection is written as Paf.DELAY, as scripts/uncalled:231-232 does."""
    RealtimeSession(unc, conf, client, pool, emit=emit, sim=sim).run(sleep=sleep)


def sim_cmd(args):
    """`uncalled sim`: the realtime loop fed from fast5 files instead of MinKNOW (scripts/uncalled:170-256)."""
    from . import _uncalled_amd as unc
    conf = unc.Conf()
    for k, v in vars(args).items():
        if v is not None and not k.startswith("_") and hasattr(conf, k) and k not in ("realtime_mode", "active_chs"):
            setattr(conf, k, v)
    conf.realtime_mode = int(unc.RealtimePool.ENRICH if args.enrich else unc.RealtimePool.DEPLETE)
    conf.active_chs = int(unc.RealtimePool.EVEN if args.even else unc.RealtimePool.ODD if args.odd else unc.RealtimePool.FULL)
    _assert_exists(conf.bwa_prefix + ".bwt")
    _assert_exists(conf.bwa_prefix + ".uncl")
    pool = None
    try:
        client = unc.ClientSim(conf)
        for f in load_fast5s(args.fast5s, args.recursive):
            if f is not None:
                client.add_fast5(f)
        client.load_fast5s()
        if not client.run():
            sys.exit(1)
        pool = unc.RealtimePool(conf)
        realtime_loop(unc, conf, client, pool, sim=True)
    except KeyboardInterrupt:
        sys.stderr.write("Keyboard interrupt\n")
    if pool is not None:
        pool.stop_all()


def get_parser():
    from . import index_params, pafstats
    d = index_params.DEFAULTS
    ap = argparse.ArgumentParser(prog="uncalled_amd", description="Rapidly maps raw nanopore signal to DNA references (MI355X)")
    sp = ap.add_subparsers(dest="subcmd")
    sp.required = True

    p = sp.add_parser("index", help="Builds the UNCALLED index of a FASTA reference")
    p.add_argument("fasta_filename", type=str, help="FASTA file to index")
    p.add_argument("-o", "--bwa-prefix", type=str, default=None, help="Index output prefix. Will use input fasta filename by default")
    p.add_argument("-s", "--max-sample-dist", type=int, default=d["max_sample_dist"], help="Maximum average sampling distance between reference self-alignments.")
    p.add_argument("--min-samples", type=int, default=d["min_samples"], help="Minimum number of self-alignments to produce (approximate, due to deterministically random start locations)")
    p.add_argument("--max-samples", type=int, default=d["max_samples"], help="Maximum number of self-alignments to produce (approximate, due to deterministically random start locations)")
    p.add_argument("-k", "--kmer-len", type=int, default=d["kmer_len"], help="Model k-mer length")
    p.add_argument("-1", "--matchpr1", type=float, default=d["matchpr1"], help="Minimum event match probability")
    p.add_argument("-2", "--matchpr2", type=float, default=d["matchpr2"], help="Maximum event match probability")
    p.add_argument("-f", "--pathlen-percentile", type=float, default=d["pathlen_percentile"], help="")
    p.add_argument("-m", "--max-replen", type=int, default=d["max_replen"], help="")
    p.add_argument("--probs", type=str, default=None, help="Find parameters with specified target probabilites (comma separated)")
    p.add_argument("--speeds", type=str, default=None, help="Find parameters with specified speed coefficents (comma separated)")
    p.add_argument("--device", type=int, default=0, help="GPU ordinal")

    p = sp.add_parser("map", help="Map fast5 files to a DNA reference")
    p.add_argument("bwa_prefix", type=str, help="BWA prefix to mapping to. Must be processed by \"uncalled index\".")
    p.add_argument("-p", "--idx-preset", type=str, default="default", help="Mapping mode")
    p.add_argument("fast5s", nargs="+", type=str, help="Reads to map: directories, fast5 files, or text files with one fast5 name per line")
    p.add_argument("-r", "--recursive", action="store_true")
    p.add_argument("-l", "--read-list", type=str, default=None, help="Only map reads with these ids")
    p.add_argument("-n", "--max-reads", type=int, default=None, help="Maximum number of reads to map")
    p.add_argument("-t", "--threads", type=int, default=1, help="1 (default): the reads are mapped exactly as `uncalled map -t 1` maps them, in input order (the few reads whose predecessor left Mapper state behind are mapped twice); "
                        "N > 1: every read independently, which is what the reference's N threads give up to their scheduling. No host threads are created either way: reads are batched onto the GPU")
    p.add_argument("--num-channels", type=int, default=512)
    p.add_argument("-e", "--max-events", type=int, default=30000, help="Will give up on a read after this many events have been processed")
    p.add_argument("-c", "--max-chunks", type=int, default=1000000, help="Will give up on a read after this many chunks have been processed")
    p.add_argument("--chunk-time", type=float, default=1, help="Length of chunks in seconds")
    p.add_argument("--device", type=int, default=0, help="GPU ordinal")
    p.add_argument("--batch-reads", type=int, default=None, help="Reads per GPU batch (default: four times what the mapper keeps in flight; the first batch is one load)")
    p.add_argument("--gpus", type=int, default=1, help="GPUs of this node to use: one worker process each, fast5 files dealt round-robin")

    p = sp.add_parser("sim", help="Simulate real-time targeted sequencing from fast5 files (enrich / deplete decisions per read)")
    p.add_argument("bwa_prefix", type=str, help="BWA prefix to mapping to. Must be processed by \"uncalled index\".")
    p.add_argument("-p", "--idx-preset", type=str, default="default", help="Mapping mode")
    p.add_argument("fast5s", nargs="+", type=str, help="Reads to replay: directories, fast5 files, or text files with one fast5 name per line")
    p.add_argument("-r", "--recursive", action="store_true")
    mode = p.add_mutually_exclusive_group(required=True)
    mode.add_argument("-D", "--deplete", action="store_true", help="Will eject reads that align to the reference")
    mode.add_argument("-E", "--enrich", action="store_true", help="Will eject reads that do not align to the reference")
    chs = p.add_mutually_exclusive_group()
    chs.add_argument("--even", action="store_true", help="Will only eject reads from even channels")
    chs.add_argument("--odd", action="store_true", help="Will only eject reads from odd channels")
    p.add_argument("-l", "--read-list", type=str, default=None, help="Only replay reads with these ids")
    p.add_argument("-n", "--max-reads", type=int, default=None, help="Maximum number of reads to replay")
    p.add_argument("--num-channels", type=int, default=512)
    p.add_argument("-e", "--max-events", type=int, default=30000)
    p.add_argument("-c", "--max-chunks", type=int, default=10, help="Will give up on a read after this many chunks have been processed")
    p.add_argument("--chunk-time", type=float, default=1, help="Length of chunks in seconds")
    p.add_argument("--duration", type=float, default=None, help="Duration to map real-time run in hours")
    p.add_argument("--device", type=int, default=0, help="GPU ordinal")

    p = sp.add_parser("pafstats", help="Computes speed and accuracy of UNCALLED mappings")
    pafstats.add_opts(p)
    return ap


def main(argv=None):
    args = get_parser().parse_args(argv)
    if args.subcmd == "index":
        index_cmd(args)
    elif args.subcmd == "map":
        map_cmd(args)
    elif args.subcmd == "sim":
        sim_cmd(args)
    else:
        from . import pafstats
        pafstats.run(args)


if __name__ == "__main__":
    main()
