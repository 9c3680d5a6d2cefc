"""Round 6: the devices= paths one by one, each in its own process with its own timeout (a hang must not cost the whole GPU slot)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = os.path.join(ROOT, "tests", "golden", "data", "MSA_RF00167_trimmed71.fa")
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl_mp.so")
HEAD = "import os, sys; sys.path.insert(0, %r)\nfrom pydca_amd.plmdca import plmdca\nf = %r\n" % (ROOT, F)
SCEN = {
    "default": "b = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0]); s = b.compute_sorted_FN_APC(); print(b.last_status['multi_gpu'], s[0])",
    "auto": "os.environ['DCA_EXCHANGE_SCHEME'] = 'auto'\nb = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0]); s = b.compute_sorted_FN_APC(); print(b.last_status['multi_gpu'], s[0])",
    "bad_device": "try:\n    plmdca.PlmDCA(f, 'rna', max_iterations=2, devices=[0, 4097]).compute_sorted_FN()\nexcept plmdca.PlmDCAException as e:\n    print('PlmDCAException', e)",
    "dies": "os.environ['DCA_MULTI_GPU_TEST_DIE_AFTER_READY'] = '1'\ntry:\n    plmdca.PlmDCA(f, 'rna', max_iterations=2, devices=[0, 0]).compute_sorted_FN()\nexcept plmdca.PlmDCAException as e:\n    print('PlmDCAException', str(e)[:200])\n"
            "del os.environ['DCA_MULTI_GPU_TEST_DIE_AFTER_READY']\nb = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0]); print('alive', b.compute_sorted_FN_APC()[0])",
}
for name in sys.argv[1:] or list(SCEN):
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, "-c", HEAD + SCEN[name]], capture_output=True, text=True, timeout=150, env=dict(os.environ, DCA_RCCL_PATH=FAKE))
        print("== %s: rc %d in %.1f s\n%s\n%s" % (name, p.returncode, time.time() - t0, p.stdout[-600:], p.stderr[-1200:]), flush=True)
    except subprocess.TimeoutExpired as e:
        print("== %s: TIMEOUT\n%s\n%s" % (name, (e.stdout or b"")[-600:], (e.stderr or b"")[-1500:]), flush=True)
