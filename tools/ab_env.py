"""CC_* environment variables -> cc_amd.config.debug (the product package itself reads no environment variable).

    import tools.ab_env; tools.ab_env.apply()

Used by bench.py, tests/conftest.py and the A/B scripts (tools/gpu.sh ab ...).  Host-glue switches are listed here; the
kernel-selection switches of the TOOLS build of the library (tools/README.md) are read by that library itself and only take
effect when CC_LIB_PATH points at tools/_bin/libccengine_tools.so."""
import os

_FLAGS = {
    "CC_NO_SLICE_GY": "no_slice_gy", "CC_NO_WGRAD_DEFER": "no_wgrad_defer", "CC_NO_SUM_N": "no_sum_n",
    "CC_NO_WGRAD_LIST": "no_wgrad_list", "CC_NO_WGRAD_QUEUE": "no_wgrad_queue",
    "CC_FORCE_COMM": "force_comm",
}


def apply(env=None):
    """-> dict of the switches that were set (for the bench line)."""
    from cc_amd import config
    env = os.environ if env is None else env
    got = {}
    for var, attr in _FLAGS.items():
        if env.get(var, "0") == "1":
            setattr(config.debug, attr, True)
            got[attr] = True
    if env.get("CC_NET_STREAMS") is not None:          # product switch cc_amd.config.net_streams (A/B: 0 / 1)
        config.net_streams = got["net_streams"] = int(env["CC_NET_STREAMS"])           # 0 / 1 (= 2 side streams) / 3
    if env.get("CC_NET_STREAM_PRIORITY"):              # e.g. "0,-1": Back2Future's stream high
        config.debug.net_stream_priority = got["net_stream_priority"] = tuple(int(v) for v in env["CC_NET_STREAM_PRIORITY"].replace(":", ",").split(","))
    if env.get("CC_GRAD_CHUNKS") is not None:          # product switch cc_amd.config.grad_chunks
        config.grad_chunks = got["grad_chunks"] = bool(int(env["CC_GRAD_CHUNKS"]))
    if env.get("CC_BIAS_TABLE") is not None:           # product switch cc_amd.config.bias_table
        config.bias_table = got["bias_table"] = bool(int(env["CC_BIAS_TABLE"]))
    if env.get("CC_LOSS_STREAM") is not None:          # product switch cc_amd.config.loss_stream
        config.loss_stream = got["loss_stream"] = bool(int(env["CC_LOSS_STREAM"]))
    if env.get("CC_CHUNK_INLINE", "0") == "1":
        config.debug.chunk_inline = got["chunk_inline"] = True
    if env.get("CC_PIPE_EXTRA"):                       # e.g. "flow:10"
        k, v = env["CC_PIPE_EXTRA"].split(":")
        config.debug.pipe_extra = got["pipe_extra"] = {k: int(v)}
    if env.get("CC_PIPE_SKIP_TAIL"):                   # measurement: networks (disp,pose,mask,flow) whose Adam segment + weight images are skipped
        config.debug.pipe_skip_tail = got["pipe_skip_tail"] = tuple(env["CC_PIPE_SKIP_TAIL"].replace(":", ",").split(","))
    if env.get("CC_PIPELINE"):                         # bench.py --pipeline default override for the `ab` step of tools/gpu.sh
        got["pipeline"] = env["CC_PIPELINE"]
    if env.get("CC_CAPTURE_MODE"):
        config.debug.capture_mode = got["capture_mode"] = env["CC_CAPTURE_MODE"]
    if env.get("CC_LIB_PATH"):
        config.debug.library_path = got["library_path"] = env["CC_LIB_PATH"]
    return got


def assert_applied():
    """The library the engine actually loaded is the one CC_LIB_PATH asked for (apply() must run before the first engine() call)."""
    from cc_amd import _lib, config
    if config.debug.library_path:
        assert os.path.samefile(_lib.engine().path, config.debug.library_path), (_lib.engine().path, config.debug.library_path)
