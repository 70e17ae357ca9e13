"""Run a command and sample rocm-smi (socket power, sclk, mclk, temperature, busy %) every 0.25 s while it runs.
usage: python tools/smi_during.py OUT.json -- <command ...>   (the program under test is a child process; nothing here touches the GPU)"""
import json, subprocess, sys, time

out = sys.argv[1]
cmd = sys.argv[sys.argv.index('--') + 1:]
p = subprocess.Popen(cmd)
samples = []
t0 = time.time()
while p.poll() is None:
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showuse', '--showtemp', '--json'], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d.get('card0', {})
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'busy', 'use', 'junction', 'edge'))}
        keep['t'] = round(time.time() - t0, 2)
        samples.append(keep)
    except Exception as e:                      # noqa
        samples.append({'t': round(time.time() - t0, 2), 'error': str(e)[:200]})
    time.sleep(0.25)
json.dump({'cmd': cmd, 'rc': p.returncode, 'samples': samples}, open(out, 'w'), indent=1)
sys.exit(p.returncode)
