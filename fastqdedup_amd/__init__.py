"""fastqdedup_amd -- MI355X-native clustering hot path of fastqdedup.

Placeholder; the host-side mirror of the reference interface is filled in
below as the HIP library lands.
"""
