"""The host blocks' own threading under ThreadSanitizer and AddressSanitizer + UBSan, on the CPU: host/jrc_blocks.cc is compiled together with
a test double of the feed ABI (tests/host_sanitize/feed_double.cc, in the place of libjrc_hip.so, which needs a GPU) and a driver that runs
the radar_chain block from a scheduler thread with idle gaps, beside the block's flusher thread and an observer calling the getters
(tests/host_sanitize/radar_chain_threads.cc).  A sanitizer report or a failed check of the driver (frame order, receive-only submissions
against the resident TX rows, nothing left in flight, one thread in the feed at a time) fails the test."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "gr-mimo-ofdm-jrc_amd", "host")
SRC = os.path.join(HERE, "host_sanitize")


BUILT = {}


def build(tmp_path, sanitize):
    if sanitize in BUILT:
        return BUILT[sanitize]
    exe = str(tmp_path / ("radar_chain_threads_" + sanitize.split(",")[0]))
    probe = subprocess.run(["g++", "-fsanitize=" + sanitize, "-x", "c++", "-", "-o", exe + "_probe"], input="int main(){return 0;}", text=True,
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("g++ -fsanitize=%s is not usable here: %s" % (sanitize, probe.stderr[-200:]))
    # the other blocks' calls into the C ABI stay unresolved: the driver only makes a radar_chain
    cmd = ["g++", "-O1", "-g", "-std=c++14", "-fsanitize=" + sanitize, "-fno-omit-frame-pointer", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
           os.path.join(SRC, "radar_chain_threads.cc"), os.path.join(SRC, "feed_double.cc"), os.path.join(HOST, "jrc_blocks.cc"),
           "-Wl,--unresolved-symbols=ignore-in-object-files", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    BUILT[sanitize] = exe
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("sanitize,max_age_us", [("thread", "2000"), ("thread", "200"), ("thread", "0"), ("address,undefined", "2000")])
def test_radar_chain_block_threads_under_sanitizers(tmp_path_factory, sanitize, max_age_us):
    exe = build(tmp_path_factory.mktemp("host_sanitize"), sanitize)
    env = dict(os.environ, JRC_RADAR_CHAIN_MAX_AGE_US=max_age_us, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="halt_on_error=1 print_stacktrace=1")
    env.pop("JRC_DEVICES", None)
    r = subprocess.run([exe, "240"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok: 240 frames"), (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("scenario,env,expect", [("destructor", {}, "ok: destructor collected"),
                                                 ("flusher_error", {"FEED_DOUBLE_FAIL_COLLECT_AT": "1", "JRC_RADAR_CHAIN_MAX_AGE_US": "8000"}, "ok: flusher error was sticky")])
def test_radar_chain_block_destructor_collects_and_flusher_errors_are_sticky(tmp_path_factory, scenario, env, expect):
    """ADVICE r4 on host/jrc_blocks.cc: a block destroyed without stop() / flush() publishes what is still in flight; a collect that fails in the
    flusher thread is said once, rethrown by the scheduler's next general_work / flush and not retried every half age bound — under
    ThreadSanitizer, against the test double of the feed ABI"""
    exe = build(tmp_path_factory.mktemp("host_sanitize"), "thread")
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", FEED_DOUBLE_LATENCY_US="1500", **env)
    e.pop("JRC_DEVICES", None)
    r = subprocess.run([exe, "0", scenario], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode == 0 and expect in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
