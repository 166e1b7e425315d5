"""CPU oracle for the YOLOv3 inference path -- TEST INFRASTRUCTURE ONLY (see y3_oracle.c)."""
