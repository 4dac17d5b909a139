import sys
sys.path.insert(0, "/root/repo")
import pocs_amd
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
for batch in (1, 16):
    for n in (2, 1000000):
        with pocs_amd.Context(0) as ctx:
            ctx.configure(plan, env, K=3, N=n, seed=1)
            ctx.set_batch(batch)
            ctx.set_option(4, 1)
            for rep in range(3):
                sys.stderr.write("batch %d N %d rep %d\n" % (batch, n, rep)); sys.stderr.flush()
                ctx.run_gmm_estimation()
            ms, l = ctx.kernel_time()
            sys.stderr.write("  event time %.2f us/launch\n" % (1e3 * ms / l))
