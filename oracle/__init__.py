"""TEST INFRASTRUCTURE ONLY. CPU restatement (torch-CPU / numpy) of the reference's arithmetic for
the hot path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (cuda-learn-notes_amd/) never does."""
from .oracle import *  # noqa: F401,F403
