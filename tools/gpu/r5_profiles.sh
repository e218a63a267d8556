#!/bin/bash
# round 5: one part of the evidence collection (PART = 1 | 2 | 3), see tools/collect_profiles.sh
R=r05 PART=${PART:-1} bash tools/collect_profiles.sh
