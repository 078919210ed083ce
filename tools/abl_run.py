import os, sys, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in [0, 1, 2, 4, 8, 3, 12, 15]:
    shutil.copy(os.path.join(root, 'tools', 'abl', f'lib_{v}.so'), os.path.join(root, 'wc_gan_amd', 'libwc_hip.so'))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fast_ablate.py')], capture_output=True, text=True).stdout
    print('ABL', v, out.strip().splitlines()[0] if out.strip() else '??', flush=True)
