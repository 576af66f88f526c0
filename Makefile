# Top-level convenience targets.
.PHONY: build test test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	$(MAKE) -C gym_uav_collision_avoidance_amd/csrc clean
	$(MAKE) -C oracle clean
